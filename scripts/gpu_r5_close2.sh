#!/bin/bash
# Round 5, final closing session at HEAD: whole GPU suite + smoke, the contract bench line exactly as the driver runs it, and
# the same command (and the opt-in split-precision one) under rocprofv3 --kernel-trace --stats.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r5_close2.sh'   ->  gpurun_out/r5close2/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5close2; mkdir -p $O; A=$PWD
echo "== gpu suite"; timeout 1200 python -m pytest tests -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
echo "== bench (contract line, default flags)"; timeout 600 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
echo "== rocprof stats, contract path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof" -o r1 --output-format csv -- python "$A/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-opt-in > "$A/$O/rocprof.log" 2>&1)
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/bench_kernel_stats.csv; head -8 "$f" | cut -c1-160; }
tail -1 $O/rocprof.log | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' '; echo
echo "== rocprof stats, split path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof_split" -o r1 --output-format csv -- python "$A/bench.py" --steps 5 --warmup 2 --split-bf16x3 --no-cpu-baseline --no-parity > "$A/$O/rocprof_split.log" 2>&1)
f=$(find $O/prof_split -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/bench_split_kernel_stats.csv; head -6 "$f" | cut -c1-160; }
rm -rf $O/prof $O/prof_split
