#!/bin/bash
# Round 5, session 1: first run of the split-precision conv (csrc/conv1d_split3.h) + its gate table, then the whole GPU
# suite and the contract bench line after the round's host-side changes.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r5_s1.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s1; mkdir -p $O
echo "== split3 tests"; timeout 420 python -m pytest tests/test_gpu_split3.py -q -m gpu --timeout 60 -s 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/tests_split3.log
echo "== split3 gate"; timeout 300 python tools/bench_split3.py --dbg --products3 --out $O/split3_gate.json 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/split3_gate.log
echo "== gpu suite"; timeout 900 python -m pytest tests -q -m gpu -x --timeout 300 --ignore tests/test_gpu_split3.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/tests_gpu.log
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 | tee $O/bench.json
