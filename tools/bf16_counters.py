#!/usr/bin/env python3
"""Counter evidence for BASELINE.json configs[4] (bf16 generator): HBM traffic, matrix-pipe busy fraction and the
shader clock the kernels actually ran at, from three rocprofv3 --pmc passes over tools/bench_decoder_bf16.py
(scripts/profile_bf16.sh):

    python tools/bf16_counters.py FETCH_DIR WRITE_DIR BUSY_DIR PASSES > profiles/rNN_bf16_counters.json

* traffic: FETCH_SIZE x read factor + WRITE_SIZE x write factor summed over every ovk16:: kernel (the bf16 convs, fused
  pairs and conv_post) of the process, divided by the number of generator passes it ran.  Counters are in KiB; the
  factors are calibrated in the same pass on three 1 GiB device-to-device copies (guide: FETCH_SIZE counts half of a
  wide streaming read on gfx950; nominal 2.0 / 1.0 when the calibration copies are missing).
* mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32); shader clock = SQ_BUSY_CYCLES / 32 /
  dispatch duration -- per kernel family and over all ovk16 kernels.  No assumed frequency anywhere.
Measurement tool."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

GIB = float(1 << 30)
FAM = re.compile(r"(conv1d_bf16cl_kernel|respair_bf16cl_kernel|respair2_bf16_kernel|conv_post_tanh_bf16\w*)<?([^>(]*)")


def rows(d):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r


def is_bf16(name):
    return "ovk16" in name or "bf16" in name


def family(name):
    m = FAM.search(name)
    return (m.group(1) + "<" + m.group(2).strip() + ">") if m else name.split("(")[0][-60:]


def traffic(d, counter):
    tot, copies = defaultdict(float), []
    for _, r in rows(d):
        if r.get("Counter_Name") != counter:
            continue
        name, v = r.get("Kernel_Name", ""), float(r.get("Counter_Value") or 0)
        if is_bf16(name):
            tot[family(name)] += v
        elif "copyBuffer" in name or "copy_kernel" in name.lower():
            copies.append(v)
    return tot, sorted(copies, reverse=True)[:3]


def main():
    fetch_dir, write_dir, busy_dir, passes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    f_tot, f_cal = traffic(fetch_dir, "FETCH_SIZE")
    w_tot, w_cal = traffic(write_dir, "WRITE_SIZE")
    rf = GIB / (sum(f_cal) / len(f_cal) * 1024) if len(f_cal) == 3 and min(f_cal) > 0 else 2.0
    wf = GIB / (sum(w_cal) / len(w_cal) * 1024) if len(w_cal) == 3 and min(w_cal) > 0 else 1.0
    per_family = {}
    for fam in sorted(set(f_tot) | set(w_tot)):
        per_family[fam] = round((rf * f_tot.get(fam, 0.0) + wf * w_tot.get(fam, 0.0)) * 1024 / passes / 1e9, 3)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (three separate passes) over "
                     "tools/bench_decoder_bf16.py --no-fp32 --steps 1 --pmc-calibration",
           "generator_passes_in_process": passes,
           "read_factor": round(rf, 3), "write_factor": round(wf, 3),
           "calibrated": len(f_cal) == 3 and len(w_cal) == 3,
           "traffic_GB_per_pass": round(sum(per_family.values()), 2),
           "traffic_GB_per_pass_by_kernel": per_family}
    per = defaultdict(lambda: defaultdict(float))
    meta = {}
    for f, r in rows(busy_dir):
        key = (f, r.get("Dispatch_Id"))
        per[key][r["Counter_Name"]] += float(r.get("Counter_Value") or 0)
        t0, t1 = r.get("Start_Timestamp"), r.get("End_Timestamp")
        meta[key] = (r.get("Kernel_Name", "?"), (int(t1) - int(t0)) if t0 and t1 else 0)
    fam = defaultdict(lambda: defaultdict(float))
    for key, ctr in per.items():
        name, ns = meta[key]
        if not is_bf16(name):
            continue
        for a in (fam[family(name)], fam["ALL ovk16 kernels"]):
            a["n"] += 1
            a["ns"] += ns
            for c, v in ctr.items():
                a[c] += v
    busy = {}
    for name, a in sorted(fam.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0)):
        cyc = a.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        if cyc <= 0:
            continue
        wave = a.get("SQ_WAVE_CYCLES", 0.0)
        busy[name] = {"launches": int(a["n"]), "ms_under_counters": round(a["ns"] / 1e6 / 1.0, 3),
                      "mfma_busy": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc), 4),
                      "shader_clock_ghz": round(cyc / a["ns"], 3) if a["ns"] else None,
                      "wait_any_frac": round(a.get("SQ_WAIT_ANY", 0.0) / wave, 3) if wave else None,
                      "lds_wait_frac": round(a.get("SQ_WAIT_INST_LDS", 0.0) / wave, 4) if wave else None}
    allk = busy.get("ALL ovk16 kernels", {})
    out["mfma_busy"] = allk.get("mfma_busy")
    out["shader_clock_ghz"] = allk.get("shader_clock_ghz")
    out["wait_any_frac"] = allk.get("wait_any_frac")
    out["note"] = ("wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES over ALL resident waves; respair2_bf16_kernel keeps four "
                   "loader waves per workgroup parked at barriers by design (they issue the LDS-DMA and the stores), so "
                   "half of its wave-cycles are waits even when its four matrix waves never stall; mfma_busy is per SIMD "
                   "and is not affected")
    out["by_kernel"] = busy
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from openvoice_amd.hostinfo import bf16_source_digest
    out["bf16_source_digest"] = bf16_source_digest()      # bench.py --bf16-generator reports the record only on a match
    out["batch"], out["frames"] = 64, 861                  # tools/bench_decoder_bf16.py defaults (BASELINE.json configs[4])
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
