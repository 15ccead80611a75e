#!/bin/bash
# kernel-trace stats of the bf16 generator at HEAD (one rocprofv3 pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
R="$PWD"; O=gpurun_out/${OUT:-r4stats}; mkdir -p $O
T="python $R/tools/bench_decoder_bf16.py --no-fp32"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d "$R/$O/prof16" -o r1 --output-format csv -- $T --steps 2 > "$R/$O/prof16.log" 2>&1)
f=$(find $O/prof16 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/bf16_kernel_stats.csv
find $O -name '*kernel_trace.csv' -size +5M -delete
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/bf16_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    n=r["Name"].replace("void ","").replace("ovk16::","").replace("ovk16p::","").replace("ovk16q::","")[:70]
    print(f"{n:72s} {int(r['Calls'])/4:5.1f}/pass {int(r['TotalDurationNs'])/4e6:7.3f} ms/pass avg {float(r['AverageNs'])/1e3:8.1f} us")
print("total ms/pass", tot/4e6)
PY
