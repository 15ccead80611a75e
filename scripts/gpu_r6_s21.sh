#!/bin/bash
# Round 6 session 21: small-launch policy of the Winograd path (engine.WINO_MIN_ITEMS): whole -m gpu suite, batch sweep.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r6_s21.sh'   ->  gpurun_out/r6s21/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s21; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 600 python tools/bench_sweep.py --no-ragged --wn-ab --batches 1 2 3 4 32 2>&1 | tee $O/batch_sweep.jsonl | cut -c1-300
