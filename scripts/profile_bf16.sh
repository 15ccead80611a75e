#!/bin/bash
# Counter evidence for BASELINE.json configs[4] (bf16 generator, batch 64) at HEAD: kernel stats, HBM FETCH / WRITE
# counters (calibrated on 1 GiB copies in the same pass) and SQ matrix-busy / clock counters, each in its OWN
# rocprofv3 pass (--kernel-trace + --pmc only), then the un-profiled bench line carrying the measured figures.
#   gpurun --timeout 900 -- 'bash scripts/profile_bf16.sh'   ->  gpurun_out/bf16_*  (copy to profiles/rNN_bf16_*)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
R="$PWD"
T="python $R/tools/bench_decoder_bf16.py --no-fp32"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof16" -o r1 --output-format csv -- $T --steps 2 > "$R/gpurun_out/prof16.log" 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/gpurun_out/pmc16_fetch" -o r1 --output-format csv -- $T --steps 1 --pmc-calibration > "$R/gpurun_out/pmc16_fetch.log" 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$R/gpurun_out/pmc16_write" -o r1 --output-format csv -- $T --steps 1 --pmc-calibration > "$R/gpurun_out/pmc16_write.log" 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY -d "$R/gpurun_out/pmc16_busy" -o r1 --output-format csv -- $T --steps 1 > "$R/gpurun_out/pmc16_busy.log" 2>&1)
python tools/bf16_counters.py gpurun_out/pmc16_fetch gpurun_out/pmc16_write gpurun_out/pmc16_busy 3 > gpurun_out/bf16_counters.json 2> gpurun_out/bf16_counters.err
python tools/pmc_summary.py gpurun_out/pmc16_fetch gpurun_out/pmc16_write > gpurun_out/bf16_pmc_fetch_write.txt 2>&1
f=$(find gpurun_out/prof16 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/bf16_kernel_stats.csv
find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
find gpurun_out/pmc16_busy gpurun_out/pmc16_fetch gpurun_out/pmc16_write -name '*.csv' -size +8M -delete
timeout 200 python tools/bench_decoder_bf16.py --steps 5 --counters gpurun_out/bf16_counters.json > gpurun_out/bf16_bench.json 2> gpurun_out/bf16_bench.err
cat gpurun_out/bf16_bench.json; head -40 gpurun_out/bf16_counters.json
