#!/bin/bash
# Round 3, session 2: length-aware work lists (tests + TTS bench with / without), bench regression check of the conv
# kernel change, bf16 power-limited MFMA ceiling microbench.  Outputs: gpurun_out/r3s2/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3s2; mkdir -p $O
echo "== limits + kernels + e2e + tts tests"; timeout 900 python -m pytest tests/test_gpu_limits.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_tts.py tests/test_gpu_torch_shim.py -q -m gpu --timeout 600 -x 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/tests.txt
echo "== bf16 MFMA ceiling"; timeout 120 tools/micro/bin/mfma_bf16_ceiling 2>&1 | tee $O/mfma_bf16_ceiling.txt
echo "== tts bench"; timeout 300 python tools/bench_tts.py --steps 10 --cpu 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_bench.json
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bench.log | cut -c1-300
