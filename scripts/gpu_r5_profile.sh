#!/bin/bash
# Round 5 profile session at HEAD: whole GPU suite, the contract bench line (carries the opt-in split measurement and the
# B = 32 CPU leg), the same command under rocprofv3 --kernel-trace --stats, the two PMC passes (FETCH_SIZE / WRITE_SIZE, each
# with the 1 GiB calibration copies), the SQ busy pass; then the same evidence for the opt-in split-precision line and the
# counter session of the opt-in bf16 generator (BASELINE.json configs[4]).
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r5_profile.sh'   ->  gpurun_out/r5prof/ (copy to profiles/r05_sNN_*)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5prof; mkdir -p $O; A=$PWD
echo "== gpu suite"; timeout 900 python -m pytest tests -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
echo "== bench (contract line)"; timeout 500 python bench.py --steps 10 --warmup 3 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
echo "== rocprof stats, contract path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof" -o r1 --output-format csv -- python "$A/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-opt-in > "$A/$O/rocprof.log" 2>&1)
grep '"metric"' $O/rocprof.log | cut -c1-200
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/bench_kernel_stats.csv; head -8 "$f" | cut -c1-160; }
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/pmc_traffic.json 2>$O/pmc_traffic.err; cat $O/pmc_traffic.json | head -30
echo "== SQ busy, contract path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$A/$O/pmc_busy" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-opt-in > "$A/$O/pmc_busy.log" 2>&1)
python tools/pmc_mfma_busy.py $O/pmc_busy > $O/pmc_mfma_busy.txt 2>&1; head -30 $O/pmc_mfma_busy.txt
# ---- opt-in split-precision line
echo "== bench --split-bf16x3"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --no-cpu-baseline 2>$O/bench_split.err | tail -1 | tee $O/bench_split.json | cut -c1-300
echo "== rocprof stats, split path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof_split" -o r1 --output-format csv -- python "$A/bench.py" --steps 5 --warmup 2 --split-bf16x3 --no-cpu-baseline --no-parity > "$A/$O/rocprof_split.log" 2>&1)
f=$(find $O/prof_split -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/bench_split_kernel_stats.csv; head -12 "$f" | cut -c1-160; }
find $O/prof_split -name '*kernel_trace.csv' -size +20M -delete
echo "== SQ busy, split path"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$A/$O/pmc_busy_split" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 1 --split-bf16x3 --no-cpu-baseline --no-parity > "$A/$O/pmc_busy_split.log" 2>&1)
python tools/pmc_mfma_busy.py $O/pmc_busy_split > $O/pmc_mfma_busy_split.txt 2>&1; head -30 $O/pmc_mfma_busy_split.txt
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
# ---- opt-in bf16 generator (configs[4]): counters, then the bench line that carries them
echo "== bf16 generator counters"; bash scripts/profile_bf16.sh > $O/profile_bf16.log 2>&1; tail -3 $O/profile_bf16.log | cut -c1-400
cp gpurun_out/bf16_counters.json $O/bf16_counters.json 2>/dev/null; cp gpurun_out/bf16_bench.json $O/bf16_bench.json 2>/dev/null; cp gpurun_out/bf16_kernel_stats.csv $O/bf16_kernel_stats.csv 2>/dev/null
mkdir -p profiles; cp $O/bf16_counters.json profiles/bf16_counters_latest.json 2>/dev/null
echo "== bench --bf16-generator --batch 64"; timeout 400 python bench.py --steps 10 --warmup 3 --bf16-generator --batch 64 --no-cpu-baseline 2>$O/bench_bf16.err | tail -1 | tee $O/bench_bf16.json | cut -c1-300
