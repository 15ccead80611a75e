#!/bin/bash
# Round 6 session 18: the whole -m gpu suite, smoke() and the contract line (default flags) with the packed-fp32 helper waves,
# the one-row-fragment Winograd instance (C = 32, k = 11, dilation 1) and the C = 64 k = 3 pairs on Winograd launches.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r6_s18.sh'   ->  gpurun_out/r6s18/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s18; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
