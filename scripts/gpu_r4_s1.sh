#!/bin/bash
# Round 4, session 1: first run of the second-generation fused bf16 pair (csrc/conv1d_bf16_pair2.hip).
#   gpurun --timeout 1200 -- 'bash scripts/gpu_r4_s1.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4s1; mkdir -p $O
echo "== pair2 tests"; timeout 400 python -m pytest tests/test_gpu_bf16_pair2.py -q -m gpu -x --timeout 120 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/tests_pair2.log
echo "== pair2 table"; timeout 300 python tools/bench_convs_bf16.py --pair2 2>&1 | grep -v amdgpu.ids | tee $O/pair2_table.log
echo "== bf16 generator"; timeout 300 python tools/bench_decoder_bf16.py --steps 5 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bf16_generator.json
echo "== bf16 tests"; timeout 400 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_pair.py -q -m gpu -x --timeout 120 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_bf16.log
