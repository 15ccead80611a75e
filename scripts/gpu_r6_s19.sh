#!/bin/bash
# Round 6 session 19: BASELINE configs[4] per stage against its binding ceiling (tools/bf16_stage_table.py), the small-batch
# sweep on the Winograd-domain path (batch 1 / 2 / 4 ... 64, WaveNet row-split A/B), and the batch-1 kernel stats.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r6_s19.sh'   ->  gpurun_out/r6s19/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r6s19; mkdir -p $O; A=$PWD
timeout 300 python tools/bf16_stage_table.py --out $O/bf16_stage_table.json 2>&1 | grep -v "^{" | tee $O/bf16_stage_table.txt
timeout 600 python tools/bench_sweep.py --no-ragged --wn-ab 2>&1 | tee $O/batch_sweep.jsonl | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof_b1" -o r1 --output-format csv -- python "$A/tools/bench_sweep.py" --batches 1 --no-ragged --steps 20 > "$A/$O/prof_b1.log" 2>&1)
f=$(find $O/prof_b1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/batch1_kernel_stats.csv; head -40 "$f" | cut -c1-160; }
find $O -name '*kernel_trace.csv' -size +20M -delete
