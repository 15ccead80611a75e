#!/bin/bash
# Round 5 closing session at HEAD: whole GPU suite + smoke, the contract bench line exactly as the driver runs it, rocprofv3
# kernel stats + SQ pass of the opt-in split-precision path (its kernels changed after scripts/gpu_r5_profile.sh ran), the
# small-batch sweep with the split path.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r5_final.sh'   ->  gpurun_out/r5final/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5final; mkdir -p $O; A=$PWD
echo "== gpu suite"; timeout 1200 python -m pytest tests -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
echo "== bench (contract line, default flags)"; timeout 600 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
echo "== rocprof stats, split path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof_split" -o r1 --output-format csv -- python "$A/bench.py" --steps 5 --warmup 2 --split-bf16x3 --no-cpu-baseline --no-parity > "$A/$O/rocprof_split.log" 2>&1)
f=$(find $O/prof_split -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/bench_split_kernel_stats.csv; head -8 "$f" | cut -c1-160; }
find $O/prof_split -name '*kernel_trace.csv' -size +20M -delete
echo "== SQ busy, split path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$A/$O/pmc_busy_split" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 1 --split-bf16x3 --no-cpu-baseline --no-parity > "$A/$O/pmc_busy_split.log" 2>&1)
python tools/pmc_mfma_busy.py $O/pmc_busy_split > $O/pmc_mfma_busy_split.txt 2>&1; head -24 $O/pmc_mfma_busy_split.txt
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
echo "== batch sweep, fp32 vs split"; timeout 600 python tools/bench_sweep.py --batches 1 2 4 8 32 --steps 5 --split-ab --no-ragged 2>&1 | grep -v amdgpu.ids | tee $O/batch_sweep_split_ab.jsonl
