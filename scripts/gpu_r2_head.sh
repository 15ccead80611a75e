#!/bin/bash
# Round 2, HEAD check: the whole -m gpu suite, smoke, then the profile session (scripts/gpu_r2_profile.sh).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
(lscpu | grep -E "Model name|^CPU\(s\)|Socket"; nproc) > gpurun_out/host_info.txt 2>&1
echo "== gpu tests"; timeout 900 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -8 | tee gpurun_out/test_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
bash scripts/gpu_r2_profile.sh
