#!/bin/bash
# Round 6 closing session at the committed sources: the whole -m gpu suite, smoke(), then scripts/gpu_r6_profile.sh (rocprofv3
# kernel stats of the contract command, the two PMC traffic passes -> pmc_traffic.json, the SQ busy pass, the contract line with
# default flags), with the PMC traffic passes of the opt-in split-precision path first (-> split3_traffic.json: the default
# line's opt_in_split_bf16x3 record reads profiles/split3_traffic_latest.json, so it is copied in place BEFORE the line is taken).
#   gpurun --timeout 3000 -- 'bash scripts/gpu_r6_close.sh'   ->  gpurun_out/r6close/ + gpurun_out/r6prof/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6close; mkdir -p $O; A=$PWD
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
echo "== split pmc fetch"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/split_pmc_fetch" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --split-bf16x3 --no-cpu-baseline --no-parity --pmc-calibration > "$A/$O/split_pmc_fetch.log" 2>&1)
echo "== split pmc write"; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$A/$O/split_pmc_write" -o r1 --output-format csv -- python "$A/bench.py" --steps 1 --warmup 0 --split-bf16x3 --no-cpu-baseline --no-parity --pmc-calibration > "$A/$O/split_pmc_write.log" 2>&1)
python tools/pmc_traffic_split.py $O/split_pmc_fetch $O/split_pmc_write 5 32 861 > $O/split3_traffic.json 2>$O/split3_traffic.err && cp $O/split3_traffic.json profiles/split3_traffic_latest.json
python -c "
import json; d=json.load(open('$O/split3_traffic.json')); print({k:v for k,v in d.items() if k!='instances'})" | cut -c1-400
find $O -name '*kernel_trace.csv' -size +20M -delete; find $O -name '*counter_collection.csv' -size +30M -delete
bash scripts/gpu_r6_profile.sh
