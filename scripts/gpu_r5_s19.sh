#!/bin/bash
# Round 5, session 19: PMC traffic record + contract line after the last edit of a digest file (dbg guard in ov_wn_layer_f32).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s19; mkdir -p $O; A=$PWD
echo "== wn tests"; timeout 300 python -m pytest tests/test_gpu_wn_layer.py -q -m gpu --timeout 200 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/tests_wn.log
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/pmc_traffic.json 2>$O/pmc_traffic.err; head -5 $O/pmc_traffic.json
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
cp $O/pmc_traffic.json profiles/pmc_traffic_latest.json
echo "== bench (contract line, default flags)"; timeout 600 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
