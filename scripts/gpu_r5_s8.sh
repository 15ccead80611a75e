#!/bin/bash
# Round 5, session 8: soak (200 steps) of the opt-in split path and of the contract path; V1 TTS (configs[3]) with the split
# generator stages; the split e2e tests incl. the fuzz cases.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s8; mkdir -p $O
echo "== split e2e tests (with fuzz cases)"; timeout 600 python -m pytest tests/test_gpu_split3_e2e.py -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests_split3_e2e.log
echo "== soak, split path, 200 steps"; timeout 400 python bench.py --steps 200 --warmup 3 --split-bf16x3 --no-cpu-baseline 2>$O/soak_split.err | tail -1 | tee $O/soak_split.json | cut -c1-220
echo "== soak, contract path, 100 steps"; timeout 400 python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-opt-in 2>$O/soak.err | tail -1 | tee $O/soak_contract.json | cut -c1-220
for m in 0 6 3; do echo "== tts, split $m, skip_padding"; timeout 300 python tools/bench_tts.py --split-bf16x3 $m 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_split$m.json | cut -c1-400; done
for m in 0 6; do echo "== tts, split $m, full padding"; timeout 300 python tools/bench_tts.py --split-bf16x3 $m --full-padding 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_full_split$m.json | cut -c1-400; done
