#!/bin/bash
# bf16 generator: parity tests, pair table (optional), generator bench, kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${OUT:-r4bf16}; mkdir -p $O
echo "== bf16 tests"; timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_pair.py tests/test_gpu_bf16_pair2.py -q -m gpu -x --timeout 200 2>&1 | grep -v amdgpu.ids | grep "passed\|failed\|rror" | tee $O/tests.log
echo "== bf16 generator"; timeout 300 python tools/bench_decoder_bf16.py --steps 5 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bf16_generator.json
OUT=${OUT:-r4bf16}/stats CALLS_DIV=4 TOP=${TOP:-50} TAIL=0 bash scripts/gpu_stats.sh python $PWD/tools/bench_decoder_bf16.py --no-fp32 --steps 2
