#!/bin/bash
# Round 5, session 20: HBM traffic of the split-precision conv launches (two PMC passes of bench.py --split-bf16x3), then
# the default bench line (its opt_in_split_bf16x3.roofline now carries the traffic) and the --split-bf16x3 line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s20; mkdir -p $O; A=$PWD
echo "== pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --split-bf16x3 --no-cpu-baseline --no-parity --pmc-calibration > "$A/$O/pmc_fetch.log" 2>&1)
echo "== pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --split-bf16x3 --no-cpu-baseline --no-parity --pmc-calibration > "$A/$O/pmc_write.log" 2>&1)
python tools/pmc_traffic_split.py $O/pmc_fetch $O/pmc_write 5 32 861 > $O/split3_traffic.json 2>$O/split3_traffic.err; python -c "
import json; d=json.load(open('$O/split3_traffic.json')); print({k:v for k,v in d.items() if k!='instances'}); [print(i) for i in d['instances']]" | cut -c1-330
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
cp $O/split3_traffic.json profiles/split3_traffic_latest.json
echo "== bench --split-bf16x3"; timeout 400 python bench.py --steps 10 --warmup 3 --split-bf16x3 --no-cpu-baseline 2>$O/bench_split.err | tail -1 | tee $O/bench_split.json | cut -c1-200
echo "== bench (contract line, default flags)"; timeout 600 python bench.py 2>$O/bench.err | tail -1 | tee $O/bench.json | cut -c1-200
