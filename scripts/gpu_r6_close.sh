#!/bin/bash
# Round 6 closing session at the committed sources: the whole -m gpu suite, smoke(), then scripts/gpu_r6_profile.sh (rocprofv3
# kernel stats of the contract command, the two PMC traffic passes -> pmc_traffic.json, the SQ busy pass, the contract line with
# default flags).   gpurun --timeout 3000 -- 'bash scripts/gpu_r6_close.sh'   ->  gpurun_out/r6close/ + gpurun_out/r6prof/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6close; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
bash scripts/gpu_r6_profile.sh
