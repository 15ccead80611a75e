#!/bin/bash
# Round 3, session 3: whole GPU suite + smoke + TTS bench at HEAD.  Outputs: gpurun_out/r3s3/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3s3; mkdir -p $O
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/gpu_tests.txt
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
echo "== tts bench"; timeout 300 python tools/bench_tts.py --steps 10 --cpu 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_bench.json
