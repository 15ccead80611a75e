#!/bin/bash
# Round 5: where a batch-1 conversion (the only batch the reference API issues) spends its time on the opt-in split path.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5b1; mkdir -p $O; A=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof" -o r1 --output-format csv -- python "$A/bench.py" --batch 1 --steps 20 --warmup 3 --split-bf16x3 --no-cpu-baseline --no-parity > "$A/$O/rocprof.log" 2>&1)
grep '"metric"' $O/rocprof.log | cut -c1-200
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/b1_split_kernel_stats.csv; head -40 "$f" | cut -c1-170; }
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
