#!/bin/bash
# Round 6 session 20: Winograd vs direct per shape at batch 1 / 2 (and the one-fragment instance, frags = 1) -- the data for
# the small-batch launch policy.   gpurun --timeout 900 -- 'bash scripts/gpu_r6_s20.sh'   ->  gpurun_out/r6s20/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s20; mkdir -p $O
for b in 1 2; do for f in 0 1; do
  timeout 300 python tools/bench_wino.py --batch $b --shapes --res --frags $f --reps 20 2>&1 | grep "^C=" | sed "s/^/B=$b frags=$f /" | cut -c1-150 | tee -a $O/wino_small_batch.txt
done; done
