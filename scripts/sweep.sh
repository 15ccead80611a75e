# Measurement helper: A/B sweep of conv kernel variants on the MRF shapes, twice (run-to-run noise on this pool is
# +-5 %).  Edit the bench_convs.py arguments for the knob under test:  gpurun -- 'bash scripts/sweep.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do
timeout 300 python tools/bench_convs.py --channels 256 128 64 32 --tpw 0 1 --reps 6 2>&1 | grep -v amdgpu.ids
done > gpurun_out/sweep.log 2>&1
tail -3 gpurun_out/sweep.log
