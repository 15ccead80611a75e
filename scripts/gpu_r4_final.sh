#!/bin/bash
# Round 4 closing session at HEAD: whole GPU suite, smoke, the contract bench exactly as the driver runs it, the fp32
# profile passes (scripts/gpu_r4_profile.sh), bf16 counters (scripts/profile_bf16.sh), small-batch sweep, TTS bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r4final}; mkdir -p $O
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | grep -v amdgpu.ids | grep "passed\|failed\|rror" | tail -6 | tee $O/gpu_tests.txt
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.log
echo "== bench (driver defaults)"; timeout 400 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bench_default.log | cut -c1-200
bash scripts/gpu_r4_profile.sh 2>&1 | tail -45
echo "== bf16"; bash scripts/profile_bf16.sh 2>&1 | tail -3 | cut -c1-900
mkdir -p $O/bf16; cp gpurun_out/bf16_* $O/bf16/ 2>/dev/null
echo "== sweep"; timeout 300 python tools/bench_sweep.py --batches 1 2 4 8 16 32 64 --steps 8 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log | cut -c1-200
echo "== tts"; timeout 300 python tools/bench_tts.py --steps 10 --cpu 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_bench.json | cut -c1-400
