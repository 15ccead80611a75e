#!/bin/bash
# Round 6 session 38: trimmed N-blocks (no halo to stage) for K = 3 and the 64- / 32-row workgroups, halo groups elsewhere: kernel tests, the
# all-shapes table, the contract line.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r6_s37.sh'   ->  gpurun_out/r6s38/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s38; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wino.py -x -q 2>&1 | tail -5 | tee $O/pytest_wino.txt
timeout 600 python tools/bench_wino.py --shapes --res --out $O/wino_table.json 2>&1 | grep -v "^{" | tee $O/wino_table.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-opt-in 2>$O/bench.err | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json
