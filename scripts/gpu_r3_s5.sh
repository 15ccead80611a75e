#!/bin/bash
# Round 3, session 5: sustained matrix-pipe ceilings (fp32 / bf16, two bf16 MFMA shapes), batch sweep + ragged batch,
# the GPU tests touched since session 3.  Outputs: gpurun_out/r3s5/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3s5; mkdir -p $O
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_torch_shim.py tests/test_gpu_limits.py -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.txt
echo "== sustained ceilings"; timeout 120 tools/micro/bin/mfma_sustained_ceilings 2>&1 | tee $O/mfma_sustained_ceilings.txt
echo "== sweep"; timeout 600 python tools/bench_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_sweep.txt
