#!/bin/bash
# Round 5, session 21: what FETCH_SIZE counts for 64-byte row segments at 128 / 256 / 512-byte stride (the split-precision
# conv's input pattern), registers vs LDS-DMA, next to a wide read of the same 1 GiB (tools/micro/fetch_size_segments.hip).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5s21; mkdir -p $O; A=$PWD
timeout 120 tools/micro/bin/fetch_size_segments | tee $O/timing.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$A/$O/pmc" -o r1 --output-format csv -- "$A/tools/micro/bin/fetch_size_segments" > "$A/$O/pmc.log" 2>&1)
python - "$O" <<'PY' | tee $O/fetch_size_by_pattern.txt
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "pmc", "**", "*counter_collection.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == "FETCH_SIZE"]
print("kernel, FETCH_SIZE KiB, x 2 x 1024 / 2^30 (1.0 = every byte counted once at the wide-read factor)")
for r in rows:
    v = float(r["Counter_Value"])
    print(f"{r['Kernel_Name'][:60]:60s} {v:14.0f} {2 * v * 1024 / 2**30:7.3f}")
PY
rm -rf $O/pmc
