#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command; prints the per-kernel table (ms per CALLS_DIV calls).
#   gpurun -- 'OUT=name CALLS_DIV=22 bash scripts/gpu_stats.sh python tools/bench_latency.py --batches 1'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
R="$PWD"; O=gpurun_out/${OUT:-stats}; mkdir -p $O
(cd /tmp && timeout ${TMO:-300} rocprofv3 --kernel-trace --stats -d "$R/$O/prof" -o r1 --output-format csv -- "$@" > "$R/$O/cmd.log" 2>&1)
grep -v amdgpu.ids $O/cmd.log | tail -${TAIL:-4}
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv
find $O -name '*kernel_trace.csv' -size +5M -delete
python - <<PY
import csv, re
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
div=float("${CALLS_DIV:-1}")
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:${TOP:-45}]:
    n=re.sub(r"\(.*","",r["Name"].replace("void ","").replace("ovk::","").replace("ovk16::","").replace("ovk16p::","").replace("ovk16q::",""))[:64]
    print(f"{n:66s} {int(r['Calls'])/div:6.1f} calls {int(r['TotalDurationNs'])/div/1e6:8.3f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
print("total kernel ms", tot/div/1e6)
PY
