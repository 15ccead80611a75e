#!/bin/bash
# Measurement only: phase timers of the MFMA conv kernel's matrix waves.  Builds an EXTRA library with OV_EXP=3 into
# /tmp on the GPU box (never over the library the package loads) and runs tools/conv_phases.py against it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -Wno-pass-failed"
lib=${EXP3_LIB:-/tmp/libopenvoice_amd_exp3.so}
[ -f "$lib" ] || make -C openvoice_amd/csrc -j32 BUILD=/tmp/ov_build_exp3 LIB=$lib CXXFLAGS="$FLAGS -DOV_EXP=3" $lib > /tmp/exp3_build.log 2>&1 || tail -5 /tmp/exp3_build.log
OPENVOICE_AMD_LIB=$lib OPENVOICE_AMD_ALLOW_EXPERIMENT=1 timeout 300 python tools/conv_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_phases.txt
