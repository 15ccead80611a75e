# Measurement only: how much of the MRF kernels' time is chunk hand-off?  Rebuilds the library on the GPU box
# with OV_EXP=1 (no staging loads, no barriers: the matrix waves free-run on stale LDS) and OV_EXP=2 (barriers
# kept, no staging loads) and times the same shapes.  Numerical results of these builds are meaningless.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for e in 0 1 2; do
  rm -rf openvoice_amd/csrc/build
  make -C openvoice_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -DOV_EXP=$e" > /dev/null 2>&1
  echo "== OV_EXP=$e"
  timeout 200 python tools/bench_convs.py --channels 128 32 --reps 6 2>&1 | grep -v amdgpu.ids | grep -v "^B=\|done\|   C "
done > gpurun_out/exp_sync.log 2>&1
tail -3 gpurun_out/exp_sync.log
