cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do
timeout 300 python tools/bench_convs.py --channels 256 128 32 --stagger 0 -1 -2 --tpw 1 2 --reps 6 2>&1 | grep -v amdgpu.ids
done > gpurun_out/sweep4.log 2>&1
tail -3 gpurun_out/sweep4.log
