#!/bin/bash
# Round 6 session 40 (final tree): batch sweep 1 / 2 / 4 / 8 / 16 / 32 (with the ragged variant), small-batch determinism soak,
# convert() file latency, bf16 per-stage table -- the secondary figures DESIGN section 8 quotes, refreshed at the closing kernels.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r6_s40.sh'   ->  gpurun_out/r6s40/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s40; mkdir -p $O
timeout 600 python tools/bench_sweep.py --batches 1 2 4 8 16 32 2>&1 | tee $O/batch_sweep.jsonl | cut -c1-260
timeout 600 python tools/soak_small_batch.py --runs 100 2>&1 | tail -3 | tee $O/soak.txt | cut -c1-300
timeout 300 python tools/bench_convert_file.py 2>&1 | tail -4 | tee $O/convert_file.txt | cut -c1-300
timeout 300 python tools/bf16_stage_table.py 2>&1 | tee $O/bf16_stage_table.txt | tail -4 | cut -c1-200
