# Measurement only: how much of the MRF kernels' time is chunk hand-off?  Builds two EXTRA libraries on the GPU box
# with OV_EXP=1 (no staging loads, no barriers: the matrix waves free-run on stale LDS) and OV_EXP=2 (barriers
# kept, no staging loads) -- into /tmp, from their own object directories, never over the library the package loads --
# and times the same shapes with each.  Numerical results of these builds are meaningless; openvoice_amd/_lib.py
# refuses to load them without OPENVOICE_AMD_ALLOW_EXPERIMENT=1.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -Wno-pass-failed"
for e in 0 1 2; do
  if [ $e = 0 ]; then lib=""; else
    lib=/tmp/libopenvoice_amd_exp$e.so
    make -C openvoice_amd/csrc -j16 BUILD=/tmp/ov_build_exp$e LIB=$lib CXXFLAGS="$FLAGS -DOV_EXP=$e" > /dev/null 2>&1
  fi
  echo "== OV_EXP=$e"
  OPENVOICE_AMD_LIB=$lib OPENVOICE_AMD_ALLOW_EXPERIMENT=1 timeout 200 python tools/bench_convs.py --channels 128 32 --reps 6 2>&1 | grep -v amdgpu.ids | grep -v "^B=\|done\|   C "
done > gpurun_out/exp_sync.log 2>&1
tail -3 gpurun_out/exp_sync.log
