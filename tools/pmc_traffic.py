#!/usr/bin/env python3
"""HBM traffic of the MRF conv launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py FETCH_DIR WRITE_DIR [STEPS_IN_PASS [BATCH [FRAMES]]] > profiles/rNN_pmc_traffic.json

STEPS_IN_PASS = conversions the profiled command ran (bench.py --steps 1 --warmup 0: the timed step, the two
PCIe-inclusive steps + their warm-up, the roofline step = 5); with it the record carries the MRF launches per step,
which bench.py compares with what it launches itself.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, other access
widths and WRITE_SIZE are uncalibrated.  The same pass therefore also runs three 1 GiB device-to-device
copies (bench.py --pmc-calibration): their known byte count gives a measured bytes-per-count factor
that is reported next to the guide's nominal factors (read x2, write x1).  Measurement tool.
"""
import csv
import glob
import json
import os
import re
import sys

# the MRF launches: single ResBlock convs (EPI = LINEAR, K in {3, 7, 11}) and the fused ResBlock pairs
MRF = re.compile(r"conv1d_mfma_kernel<(3|7|11), (1|3|5), \d+, \d+, \d+, \d+, \d+, (?:true|1|2), 0, \d+>|respair_mfma_kernel<|"
                 r"conv1d_wino_kernel<")
GIB = float(1 << 30)


def read(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    rows.append((r.get("Kernel_Name", ""), float(r.get("Counter_Value") or 0),
                                 int(r.get("Grid_Size") or 0)))
    return rows


def summarise(rows, floor_kib):
    # conv_pre (k7 at frame rate) is the same template as the MRF convs and, now that large launches are
    # persistent, grid size no longer tells them apart; traffic does: conv_pre moves 21 MB in / 56 MB out per batch,
    # the smallest MRF launch (stage 0) 226 MB each way, so a per-counter floor separates them.
    mrf = [v for n, v, g in rows if MRF.search(n) and v >= floor_kib]
    # the calibration copies: the three largest launches of a copy kernel
    copies = sorted((v for n, v, g in rows if "copyBuffer" in n or "copy_kernel" in n.lower()), reverse=True)[:3]
    return mrf, copies


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    f_mrf, f_cal = summarise(read(fetch_dir, "FETCH_SIZE"), 80000.0)     # raw FETCH_SIZE is half the bytes
    w_mrf, w_cal = summarise(read(write_dir, "WRITE_SIZE"), 100000.0)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openvoice_amd.engine import PAIR_POLICY
    from openvoice_amd.hostinfo import kernel_source_digest, launch_config_digest
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    frames = int(sys.argv[5]) if len(sys.argv) > 5 else 861
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), counters in KiB",
           "kernel_source_digest": kernel_source_digest(),
           "launch_config_digest": launch_config_digest(batch, frames, True, PAIR_POLICY),
           "launch_config": {"batch_per_gpu": batch, "frames": frames, "fuse_pairs": True,
                             "pair_policy": sorted(f"C={c} k={k}" for c, k in PAIR_POLICY)},
           "steps_in_pass": steps,
           "launches_per_step": (len(f_mrf) / steps if len(f_mrf) % steps == 0 else None),
           "mrf_launches_fetch_pass": len(f_mrf), "mrf_launches_write_pass": len(w_mrf)}
    if f_mrf and w_mrf:
        fetch_kib = sum(f_mrf) / len(f_mrf)
        write_kib = sum(w_mrf) / len(w_mrf)
        out["fetch_size_kib_per_launch"] = round(fetch_kib, 1)
        out["write_size_kib_per_launch"] = round(write_kib, 1)
        out["nominal"] = {"read_factor": 2.0, "write_factor": 1.0,
                          "bytes_per_launch": round((2.0 * fetch_kib + write_kib) * 1024)}
        if f_cal and w_cal:
            rf = GIB / (sum(f_cal) / len(f_cal) * 1024)   # true bytes per counted byte, 1 GiB copy
            wf = GIB / (sum(w_cal) / len(w_cal) * 1024)
            out["calibrated"] = {"read_factor": round(rf, 3), "write_factor": round(wf, 3),
                                 "calibration": "3 x 1 GiB device-to-device copy in the same pass",
                                 "bytes_per_launch": round((rf * fetch_kib + wf * write_kib) * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
