#!/bin/bash
# Round 5, session 25: the other configurations of BASELINE.json at the final tree: V1 TTS (configs[3]: fp32, split 6 / 3
# products) and the bf16 generator (configs[4], batch 64) -- records only, their kernels did not change since r05 s4 / s10.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s25; mkdir -p $O
for m in 0 6 3; do echo "== tts, split $m"; timeout 300 python tools/bench_tts.py --split-bf16x3 $m 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_split$m.json | cut -c1-300; done
echo "== bf16 generator, batch 64"; timeout 300 python tools/bench_decoder_bf16.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/bf16_generator_batch64.json | cut -c1-300
echo "== bench --bf16-generator --batch 64"; timeout 300 python bench.py --bf16-generator --batch 64 --steps 5 --warmup 2 --no-cpu-baseline 2>$O/bench_bf16.err | tail -1 | tee $O/bench_opt_in_bf16_generator_batch64.json | cut -c1-300
