#!/bin/bash
# Measurement only: what do the weight stream / the LDS operand reads / the MFMAs themselves cost the fused pair in
# time AND in clock (power)?  Runs tools/pair2_phases.py with the production library and with the OV_EXP = 10 .. 13
# builds of scripts/build_exp_pair2.sh (built in the build container, shipped with the snapshot).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/${OUT:-exp_pair2}; mkdir -p $O
for e in 0 ${EXPS:-10 11 12 13}; do
  if [ $e = 0 ]; then lib=""; else lib=$PWD/openvoice_amd/csrc/build_exp$e/libopenvoice_amd_exp$e.so; fi
  echo "== OV_EXP=$e"
  OPENVOICE_AMD_LIB=$lib OPENVOICE_AMD_ALLOW_EXPERIMENT=1 P2_CASES="${P2_CASES:-11,1,0;3,1,0}" timeout 200 python tools/pair2_phases.py ${CH:-128} 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/exp_pair2.log | grep "^==\|^C="
