#!/bin/bash
# bf16 generator loop: parity tests, phase timers, per-shape table, decoder bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/bf16; mkdir -p $O
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_bf16_pair.py -q -m gpu -x --timeout 200 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.log
echo "== phases"; timeout 300 python tools/conv_bf16_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/phases.txt | head -${PH_HEAD:-16}
echo "== convs"; timeout 300 python tools/bench_convs_bf16.py --reps 4 ${CONVS:-} 2>&1 | grep -v amdgpu.ids | tee $O/convs.log | tail -45
echo "== decoder"; timeout 300 python tools/bench_decoder_bf16.py 2>&1 | grep -v amdgpu.ids | tee $O/decoder.log | tail -8
