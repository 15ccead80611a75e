# rocprofv3 evidence for BASELINE.json configs[4] (bf16 generator, batch 64): kernel stats + HBM counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof16" -o r1 --output-format csv -- python "$OLDPWD/tools/bench_decoder_bf16.py" --no-fp32 --steps 2 > "$OLDPWD/gpurun_out/prof16.log" 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc16_fetch" -o r1 --output-format csv -- python "$OLDPWD/tools/bench_decoder_bf16.py" --no-fp32 --steps 1 > "$OLDPWD/gpurun_out/pmc16_fetch.log" 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OLDPWD/gpurun_out/pmc16_write" -o r1 --output-format csv -- python "$OLDPWD/tools/bench_decoder_bf16.py" --no-fp32 --steps 1 > "$OLDPWD/gpurun_out/pmc16_write.log" 2>&1)
find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
python tools/pmc_summary.py gpurun_out/pmc16_fetch gpurun_out/pmc16_write > gpurun_out/pmc16_summary.txt 2>&1
grep workload gpurun_out/prof16.log | cut -c1-300
head -8 gpurun_out/prof16/r1_kernel_stats.csv
