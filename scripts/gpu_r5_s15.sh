#!/bin/bash
# Round 5, session 15: kernel stats of a batch-1 conversion with the WaveNet row-split pair.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT15:-r5s15}; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b1 --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_sweep.py --batches 1 --steps 20 --no-ragged > $GRAFT_REPO_ROOT/$O/sweep.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200; cp "$f" $O/batch1_kernel_stats.csv
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# gaps around the wn_layer kernels in the last conversion
wn = [i for i, r in enumerate(rows) if "wn_layer_kernel" in r["Kernel_Name"]]
last = wn[-96:]
d1 = [int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in last if ", 1>" in rows[i]["Kernel_Name"]]
d2 = [int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) for i in last if ", 2>" in rows[i]["Kernel_Name"]]
gaps = [int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) for i in last]
print("gate launch us", sum(d1) / max(1, len(d1)) / 1e3, "res/skip launch us", sum(d2) / max(1, len(d2)) / 1e3, "gap before each us", sum(gaps) / len(gaps) / 1e3, len(d1), len(d2))
PY
rm -rf $O/prof
