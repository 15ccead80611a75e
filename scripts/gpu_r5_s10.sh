#!/bin/bash
# Round 5, session 10: length-aware work lists in the split-precision conv (col_limit): kernel tests, the e2e fuzz cases with
# skip_padding, the ragged batch with / without, V1 TTS; then the PMC traffic passes of the contract path again (the C ABI
# header and engine.py are part of its launch-configuration digest).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s10; mkdir -p $O; A=$PWD
echo "== split3 tests"; timeout 600 python -m pytest tests/test_gpu_split3.py tests/test_gpu_split3_e2e.py tests/test_gpu_limits.py -q -m gpu --timeout 300 -x 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.log
echo "== ragged batch"; timeout 400 python tools/bench_sweep.py --batches 32 --steps 5 --split-ab 2>&1 | grep -v amdgpu.ids | tee $O/ragged_split.jsonl
for m in 6 3; do echo "== tts, split $m, skip_padding"; timeout 300 python tools/bench_tts.py --split-bf16x3 $m 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/tts_split$m.json | cut -c1-330; done
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/pmc_traffic.json 2>$O/pmc_traffic.err; head -12 $O/pmc_traffic.json
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
