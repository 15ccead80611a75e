#!/bin/bash
# Round 6 profile session at the committed sources: rocprofv3 kernel stats of the contract command, the two PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE, separate, with the 1 GiB calibration copies) -> profiles/pmc_traffic_latest.json, the SQ
# busy pass, and the contract line with default flags.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_r6_profile.sh'   ->  gpurun_out/r6prof/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6prof; mkdir -p $O; A=$PWD
echo "== rocprof stats"; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$A/$O/prof" -o r1 --output-format csv -- python "$A/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-opt-in > "$A/$O/rocprof.log" 2>&1)
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { cp "$f" $O/bench_kernel_stats.csv; head -12 "$f" | cut -c1-170; }
tail -1 $O/rocprof.log | cut -c1-200 > $O/bench_line_under_rocprof.head
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/pmc_traffic.json 2>$O/pmc_traffic.err && cp $O/pmc_traffic.json profiles/pmc_traffic_latest.json; head -12 $O/pmc_traffic.json
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
echo "== SQ busy"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$A/$O/pmc_busy" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-opt-in > "$A/$O/pmc_busy.log" 2>&1)
python tools/pmc_mfma_busy.py $O/pmc_busy > $O/pmc_mfma_busy.txt 2>&1; head -30 $O/pmc_mfma_busy.txt
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
echo "== bench (contract line, default flags)"; timeout 900 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
