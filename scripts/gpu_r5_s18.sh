#!/bin/bash
# Round 5, session 18: after the deep-prefetch gate launch of the WaveNet row-split pair: whole GPU suite, smoke, the PMC traffic
# passes of the contract path again (header + engine.py are part of its launch-configuration digest), the contract bench
# line with the driver's defaults, the small-batch sweep with the row-split A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s18; mkdir -p $O; A=$PWD
echo "== gpu suite"; timeout 1200 python -m pytest tests -q -m gpu --timeout 300 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/tests_gpu.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/smoke.log
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-opt-in --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/pmc_traffic.json 2>$O/pmc_traffic.err; head -12 $O/pmc_traffic.json
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
cp $O/pmc_traffic.json profiles/pmc_traffic_latest.json
echo "== bench (contract line, default flags)"; timeout 600 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-300
echo "== sweep, WaveNet row split A/B"; timeout 600 python tools/bench_sweep.py --batches 1 2 3 4 32 --steps 30 --wn-ab --split-ab --no-ragged 2>$O/sweep.err | tee $O/batch_sweep_wn_row_split_ab.jsonl | cut -c1-600
