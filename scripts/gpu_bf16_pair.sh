#!/bin/bash
# bf16 fused-pair kernel: parity tests, per-shape table, generator bench.  Outputs: gpurun_out/bf16pair/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/bf16pair; mkdir -p $O
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_bf16_pair.py tests/test_gpu_bf16.py -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.txt
echo "== pair table"; timeout 300 python tools/bench_convs_bf16.py --pair 2>&1 | grep -v amdgpu.ids | tail -14 | tee $O/pair_table.txt
echo "== generator"; timeout 300 python tools/bench_decoder_bf16.py --no-fp32 --steps 10 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/generator.json
