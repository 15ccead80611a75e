#!/bin/bash
# Round 5, session 13: four-set weight ring (requests two pairs ahead) in the C = 64 geometry of the split-precision conv.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s13; mkdir -p $O
echo "== split3 tests"; timeout 600 python -m pytest tests/test_gpu_split3.py tests/test_gpu_split3_e2e.py -q -m gpu --timeout 300 -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log
echo "== split3 table"; timeout 400 python tools/bench_split3.py --shapes --reps 6 --dbg --out $O/split3_table.json 2>&1 | grep -v amdgpu.ids | grep -v "^{" | grep "C=64\|C=128 K=11 d=1" | cut -c1-330 | tee $O/split3_table.log
echo "== bench, opt-in split line"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --no-cpu-baseline 2>$O/bench_split.err | tail -1 | tee $O/bench_split.json | cut -c1-200
echo "== bench, opt-in split line, 3 products"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --split-products 3 --no-cpu-baseline 2>$O/bench_split3p.err | tail -1 | tee $O/bench_split_3products.json | cut -c1-200
