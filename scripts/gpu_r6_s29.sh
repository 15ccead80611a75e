#!/bin/bash
# Round 6 session 29: three-pair weight ring (requests two k-step pairs ahead) where a chunk is a whole number of revolutions (K = 7):
# ds_write2_b32) and the one-row-fragment workgroup (C = 32, K = 11): kernel tests, phase timers, the all-shapes table,
# the contract line, and the contract line with the C = 64 k = 3 pairs on Winograd launches instead of fused direct pairs.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r6_s16.sh'   ->  gpurun_out/r6s29/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -5 | tee $O/pytest_wino.txt
for a in "--C 128 --K 3" "--C 128 --K 7" "--C 128 --K 11" "--C 128 --K 3 --res" "--C 64 --K 11" "--C 64 --K 3" "--C 32 --K 11" "--C 32 --K 11 --dil 5" "--C 128 --K 11 --dil 5"; do
  timeout 120 python tools/wino_phases.py $a 2>&1 | tail -1 | tee -a $O/phases.jsonl; done
timeout 600 python tools/bench_wino.py --shapes --res --out $O/wino_table.json 2>&1 | grep -v "^{" | tee $O/wino_table.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json
timeout 300 python -c "
import sys, runpy
import openvoice_amd.engine as e
e.PAIR_POLICY = {(32, 3), (32, 7)}
sys.argv = ['bench.py', '--steps', '10', '--warmup', '3', '--no-cpu-baseline', '--no-opt-in']
runpy.run_path('bench.py', run_name='__main__')" 2>$O/bench_c64k3_wino.err | tail -1 > $O/bench_c64k3_wino.json; cut -c1-400 $O/bench_c64k3_wino.json
