#!/bin/bash
# Round 5, session 3: batch sweep, fp32 path vs the opt-in split-precision MRF (6 and 3 products).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s3; mkdir -p $O
timeout 600 python tools/bench_sweep.py --batches 1 2 4 8 16 32 64 --steps 5 --split-ab --no-ragged 2>&1 | grep -v amdgpu.ids | tee $O/batch_sweep_split_ab.jsonl
