# One rocprofv3 counter pass (SQ counters only, with --kernel-trace; no other trace domains) over one bench step:
# matrix-pipe busy fraction per kernel family at the clock the kernels actually ran at.
#   gpurun --timeout 300 -- 'bash scripts/profile_mfma_busy.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/pmc_busy
(cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY \
   -d "$OLDPWD/gpurun_out/pmc_busy" -o r1 --output-format csv -- python "$OLDPWD/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/pmc_busy.log" 2>&1)
python tools/pmc_mfma_busy.py gpurun_out/pmc_busy > gpurun_out/pmc_mfma_busy.txt 2>&1
cat gpurun_out/pmc_mfma_busy.txt
head -1 $(find gpurun_out/pmc_busy -name '*counter_collection.csv' | head -1) > gpurun_out/pmc_busy_header.txt
find gpurun_out/pmc_busy -name '*.csv' -size +8M -delete
