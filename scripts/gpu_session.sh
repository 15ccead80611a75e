#!/bin/bash
# One gpurun call = tests + bench + rocprof, everything logged under gpurun_out/.
#   gpurun --timeout 900 -- 'bash scripts/gpu_session.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
echo "== kernel tests" ; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout 300 2>&1 | tail -25 | tee gpurun_out/test_kernels.log
echo "== e2e tests" ; timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s --timeout 600 2>&1 | tail -25 | tee gpurun_out/test_e2e.log
echo "== bench" ; timeout 600 python bench.py --steps 3 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench.log
