#!/bin/bash
# Round 5, session 2: split-precision conv generalised to C = 64, the engine's opt-in split path end to end (parity at
# the fp32 bars), and the opt-in bench line next to the contract line.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r5_s2.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s2; mkdir -p $O
echo "== split3 tests"; timeout 420 python -m pytest tests/test_gpu_split3.py -q -m gpu --timeout 60 -x 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/tests_split3.log
echo "== split3 e2e tests"; timeout 600 python -m pytest tests/test_gpu_split3_e2e.py -q -m gpu --timeout 300 -s 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/tests_split3_e2e.log
echo "== bench, opt-in split line"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --no-cpu-baseline 2>$O/bench_split.err | tail -1 | tee $O/bench_split.json
echo "== bench, opt-in split line, 3 products"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --split-products 3 --no-cpu-baseline 2>$O/bench_split3p.err | tail -1 | tee $O/bench_split_3products.json
echo "== bench, contract line (carries opt_in_split_bf16x3)"; timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-batch32 2>$O/bench.err | tail -1 | tee $O/bench.json
echo "== split3 table"; timeout 400 python tools/bench_split3.py --shapes --reps 6 --out $O/split3_table.json 2>&1 | grep -v amdgpu.ids | grep -v "^{" | tee $O/split3_table.log
