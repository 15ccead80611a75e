#!/bin/bash
# Measurement variants of the Winograd-domain conv kernel (csrc/conv1d_wino.h, OVW_EXP): only conv1d_wino.hip is
# recompiled, the other objects are linked from the production build.  Usage: scripts/exp_wino.sh <n>  ->
# openvoice_amd/libopenvoice_amd_wexp<n>.so (select it with OPENVOICE_AMD_LIB=...; ctypes binding).  Results of these
# builds are meaningless; they exist to time phases.
set -e
cd "$(dirname "$0")/../openvoice_amd/csrc"
n=${1:?variant}
make -j8 >/dev/null
mkdir -p build_wexp$n
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-result -Wno-pass-failed \
  -DOVW_EXP=$n -c conv1d_wino_k11.hip -o build_wexp$n/conv1d_wino_k11.o
objs=$(ls build/*.o | grep -v conv1d_wino_k11)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build_wexp$n/conv1d_wino_k11.o -o ../libopenvoice_amd_wexp$n.so
echo ../libopenvoice_amd_wexp$n.so
