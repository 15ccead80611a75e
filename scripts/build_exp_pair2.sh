#!/bin/bash
# Measurement builds of the second-generation bf16 pair (conv1d_bf16_pair2.hip, OV_EXP = 10 .. 13): cross-compiled HERE
# into openvoice_amd/csrc/build_exp<N>/libopenvoice_amd_exp<N>.so (they travel to the GPU box with the snapshot); only
# the pair kernel and ov_api.hip (ov_build_experiment) are rebuilt, every other object is the production one.
#   bash scripts/build_exp_pair2.sh 10 11 12 13
set -e
cd "$(dirname "$0")/../openvoice_amd/csrc"
make -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -Wno-pass-failed"
for e in "$@"; do
  d=build_exp$e; mkdir -p $d
  for f in conv1d_bf16_pair2 conv1d_bf16_pair2_k3 conv1d_bf16_pair2_k7 conv1d_bf16_pair2_k11 ov_api; do
    /opt/rocm/bin/hipcc $FLAGS -DOV_EXP=$e -c $f.hip -o $d/$f.o &
  done
  wait
  objs=$(ls build/*.o | grep -v "conv1d_bf16_pair2\|ov_api.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $d/conv1d_bf16_pair2*.o $d/ov_api.o -o $d/libopenvoice_amd_exp$e.so
  echo "built $d/libopenvoice_amd_exp$e.so"
done
