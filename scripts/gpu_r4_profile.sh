#!/bin/bash
# Round 4 profile session at HEAD: contract bench line, the same command under rocprofv3 --kernel-trace --stats, the two
# PMC passes (FETCH_SIZE / WRITE_SIZE, each with the 1 GiB calibration copies), the SQ busy pass, the forced-dist line.
# Outputs: gpurun_out/r4prof/ (copy to profiles/r03_sNN_*).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4prof; mkdir -p $O; A=$PWD
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 3 --cpu-budget 12 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bench.log | cut -c1-300
echo "== bench --force-dist (RCCL at world 1)"; timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --force-dist 2>&1 | grep '"metric"' | tee $O/bench_force_dist.log | cut -c1-420
echo "== rocprof stats"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$A/$O/prof" -o r1 --output-format csv -- python "$A/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-parity > "$A/$O/rocprof.log" 2>&1)
grep '"metric"' $O/rocprof.log | cut -c1-200
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/bench_kernel_stats.csv; head -8 "$f" | cut -c1-160; }
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/pmc_traffic.json 2>$O/pmc_traffic.err; cat $O/pmc_traffic.json
echo "== SQ busy"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$A/$O/pmc_busy" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-parity > "$A/$O/pmc_busy.log" 2>&1)
python tools/pmc_mfma_busy.py $O/pmc_busy > $O/pmc_mfma_busy.txt 2>&1; head -40 $O/pmc_mfma_busy.txt
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
