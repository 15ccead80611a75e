#!/bin/bash
# Round 4, session 2+: phase timers of the second-generation fused pair (+ parity, + table when asked).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r4s2}; mkdir -p $O
echo "== pair2 tests"; timeout 400 python -m pytest tests/test_gpu_bf16_pair2.py -q -m gpu -x --timeout 120 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests_pair2.log
echo "== pair2 phases"; timeout 300 python tools/pair2_phases.py ${PHASES_ARGS:-} 2>&1 | grep -v amdgpu.ids | tee $O/pair2_phases.log
if [ "${TABLE:-0}" = "1" ]; then
  echo "== pair2 table"; timeout 300 python tools/bench_convs_bf16.py --pair2 2>&1 | grep -v amdgpu.ids | tee $O/pair2_table.log
  echo "== bf16 generator"; timeout 300 python tools/bench_decoder_bf16.py --steps 5 --no-fp32 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/bf16_generator.json
fi
