#!/bin/bash
# Round 5, session 14: row-split launch pair of the WaveNet layer for one or two utterances (wn_layer.hip MODE 1 + 2).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s14; mkdir -p $O
echo "== wn layer tests"; timeout 600 python -m pytest tests/test_gpu_wn_layer.py -q -m gpu --timeout 300 -x 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_wn.log
echo "== e2e tests"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_tts.py tests/test_gpu_limits.py -q -m gpu --timeout 600 -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests_e2e.log
echo "== sweep, WaveNet row split A/B"; timeout 600 python tools/bench_sweep.py --batches 1 2 3 4 --steps 20 --wn-ab --split-ab --no-ragged 2>$O/sweep.err | tee $O/batch_sweep_wn_row_split_ab.jsonl | cut -c1-600
