#!/bin/bash
# One gpurun call = tests + bench + rocprof, everything logged under gpurun_out/.
#   gpurun --timeout 900 -- 'bash scripts/gpu_session.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu_info.txt
echo "== kernel tests" ; timeout 400 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout 120 2>&1 | tail -25 | tee gpurun_out/test_kernels.log
echo "== e2e tests" ; timeout 400 python -m pytest tests/test_gpu_e2e.py -q -m gpu -s --timeout 300 2>&1 | tail -25 | tee gpurun_out/test_e2e.log
echo "== bench" ; timeout 300 python bench.py --steps 3 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench.log
echo "== rocprof" ; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r1 --output-format csv -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1)
tail -3 gpurun_out/rocprof.log
find gpurun_out/prof -name '*stats*' | head; f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -30 "$f"
# keep the merged-back payload small: the per-dispatch trace can be tens of MB
find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete
