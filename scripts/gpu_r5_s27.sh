#!/bin/bash
# Round 5, session 27: device resampler at the audio boundary (ov_polyphase_fir_f32, C ABI 2.06): its tests, the e2e tests
# that go through convert / extract_se, the file-level latency of convert() -- then, because the C ABI header is part of the
# contract path's launch-configuration digest, the PMC traffic passes again and the contract line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s27; mkdir -p $O; A=$PWD
echo "== resampler + e2e tests"; timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_e2e.py -q -m gpu --timeout 600 -x 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/tests.log
echo "== convert() on files"; timeout 500 python tools/bench_convert_file.py --runs 20 2>$O/convert_file.err | tee $O/convert_file_latency.jsonl | cut -c1-500
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/pmc_traffic.json 2>$O/pmc_traffic.err; head -5 $O/pmc_traffic.json
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
cp $O/pmc_traffic.json profiles/pmc_traffic_latest.json
echo "== bench (contract line, default flags)"; timeout 600 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
