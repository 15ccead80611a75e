cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for rep in 1 2; do
timeout 300 python tools/bench_convs.py --channels 256 128 --kernels 7 11 --tiles 1 --loaders 2 --chunks 16 32 --reps 6 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/bench_convs.py --channels 64 --kernels 7 11 --tiles 2 --loaders 4 --chunks 16 32 --reps 6 2>&1 | grep -v amdgpu.ids
done > gpurun_out/sweep2.log 2>&1
tail -5 gpurun_out/sweep2.log
