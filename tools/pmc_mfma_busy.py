#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel family from one rocprofv3 --pmc pass over bench.py.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS ... -- python bench.py ...
    python tools/pmc_mfma_busy.py DIR > profiles/rNN_pmc_mfma_busy.txt

SQ_VALU_MFMA_BUSY_CYCLES sums, over the 1024 SIMDs, the cycles a matrix instruction occupies the pipe (64 per
v_mfma_f32_32x32x2_f32: checked against the instruction count); SQ_BUSY_CYCLES is summed over the 32 shader engines
(8 XCDs x 4), so SQ_BUSY_CYCLES / 32 is the kernel's length in shader clocks and

    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * SQ_BUSY_CYCLES / 32)

is the fraction of matrix-pipe cycles in use AT THE CLOCK THE KERNEL ACTUALLY RAN AT (no assumed frequency).
Where the CSV carries dispatch timestamps the implied shader clock (SQ_BUSY_CYCLES / 32 / duration) is printed too.
Kernels run serialised under counter collection, so durations here are not the bench's.  Measurement tool."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

CONV = re.compile(r"conv1d_mfma_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false|\d+), (\d+), (\d+)>")   # staging kind: bool before r01 s40, int after
EPI = {0: "linear", 1: "gate", 2: "resskip", 3: "couple", 4: "posterior", 5: "convT", 6: "magnitude", 16: "convT s8",
       17: "convT s2"}


WINO = re.compile(r"conv1d_wino_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false|0|1)>")


def family(name):
    w = WINO.search(name)
    if w:
        k, dil, ci, nf, mw = (int(x) for x in w.groups()[:5])
        return f"winograd k={k} d={dil} rows {32 * mw} frags {nf} chunk {ci}"
    m = CONV.search(name)
    if not m:
        return name.split("(")[0][-60:]
    k, d, wm, wn, wvm, wvn, chunk, stage, epi, nld = (int(x) if x.isdigit() else x for x in m.groups())
    tile = f"{32 * wm * wvm}x{32 * wn * wvn}"
    return f"conv k={k} {EPI.get(epi, epi)} tile {tile} chunk {chunk} nld {nld}"


def main():
    d = sys.argv[1]
    per = defaultdict(lambda: defaultdict(float))     # dispatch -> counter -> value
    meta = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                key = (f, r.get("Dispatch_Id"))
                per[key][r["Counter_Name"]] += float(r.get("Counter_Value") or 0)
                t0, t1 = r.get("Start_Timestamp"), r.get("End_Timestamp")
                meta[key] = (r.get("Kernel_Name", "?"), (int(t1) - int(t0)) if t0 and t1 else 0)
    fam = defaultdict(lambda: defaultdict(float))
    for key, ctr in per.items():
        name, ns = meta[key]
        a = fam[family(name)]
        a["n"] += 1
        a["ns"] += ns
        for c, v in ctr.items():
            a[c] += v
    rows = sorted(fam.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0))
    tot = defaultdict(float)
    print(f"{'kernel family':58s} {'n':>4s} {'Mcycles':>9s} {'mfma_busy':>9s} {'lds_wait':>8s} {'clock GHz':>9s}")
    for name, a in rows:
        cyc = a.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        if cyc <= 0:
            continue
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc)
        wave = a.get("SQ_WAVE_CYCLES", 0.0)
        lds = a.get("SQ_WAIT_INST_LDS", 0.0) / wave if wave else float("nan")
        ghz = cyc / a["ns"] if a["ns"] else float("nan")
        for c in ("SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"):
            tot[c] += a.get(c, 0.0)
        tot["ns"] += a["ns"]
        if cyc >= 1e5:
            print(f"{name:58s} {int(a['n']):4d} {cyc / 1e6:9.2f} {busy:9.3f} {lds:8.3f} {ghz:9.2f}")
    cyc = tot["SQ_BUSY_CYCLES"] / 32.0
    if cyc:
        print(f"{'ALL KERNELS':58s} {'':4s} {cyc / 1e6:9.2f} {tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):9.3f} {'':8s} "
              f"{(cyc / tot['ns']) if tot['ns'] else float('nan'):9.2f}")


if __name__ == "__main__":
    main()
