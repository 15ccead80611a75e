"""The rocprofv3 post-processing tools (measurement infrastructure behind profiles/ and bench.py's `traffic`) on
synthetic counter CSVs with hand-computable answers.  CPU only."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
MRF = "void ovk::conv1d_mfma_kernel<11, 1, 2, 2, 2, 2, 32, true, 0, 2>(ov_conv1d_params)"
PRE = "void ovk::conv1d_mfma_kernel<7, 1, 2, 2, 2, 2, 32, 1, 0, 2>(ov_conv1d_params)"      # name after r01 s40 (int staging kind)
COPY = "__amd_rocclr_copyBuffer"


def _write(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(HDR)
        w.writerows(rows)


def _run(tool, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], capture_output=True, text=True,
                         check=True)
    return out.stdout


def test_mfma_busy_fraction_and_clock_from_sq_counters(tmp_path):
    # one kernel: 155 200 MFMAs x 64 cycles on the pipe, 10 M shader cycles long (SQ_BUSY_CYCLES is summed over the
    # 32 shader engines), 4.0 ms dispatch -> busy = 155200 * 64 / (1024 * 1e7) = 0.00097; clock = 1e7 / 4e6 ns = 2.5 GHz
    rows = [[1, 512, MRF, "SQ_VALU_MFMA_BUSY_CYCLES", 0.9 * 1024 * 1e7, 1000, 4001000],
            [1, 512, MRF, "SQ_BUSY_CYCLES", 32 * 1e7, 1000, 4001000],
            [1, 512, MRF, "SQ_WAVE_CYCLES", 4e9, 1000, 4001000],
            [1, 512, MRF, "SQ_WAIT_INST_LDS", 4e7, 1000, 4001000]]
    _write(str(tmp_path / "p" / "r1_counter_collection.csv"), rows)
    out = _run("pmc_mfma_busy.py", str(tmp_path / "p"))
    line = [l for l in out.splitlines() if l.startswith("conv k=11 linear tile 128x128 chunk 32 nld 2")][0].split()
    assert line[-3] == "0.900" and line[-2] == "0.010" and line[-1] == "2.50"
    assert "ALL KERNELS" in out


def test_pmc_traffic_applies_guide_factors_and_separates_conv_pre(tmp_path):
    # MRF launch: FETCH_SIZE 500 000 KiB (x2 per the guide's gfx950 correction), WRITE_SIZE 900 000 KiB;
    # conv_pre (same template, tiny traffic) must not dilute the mean; 1 GiB calibration copies: fetch counts half.
    gib_kib = float(1 << 20)
    fetch = [[1, 512, MRF, "FETCH_SIZE", 500000.0, 0, 1], [2, 512, MRF, "FETCH_SIZE", 500000.0, 0, 1],
             [3, 512, PRE, "FETCH_SIZE", 10000.0, 0, 1]] + [[10 + i, 1, COPY, "FETCH_SIZE", gib_kib / 2, 0, 1] for i in range(3)]
    write = [[1, 512, MRF, "WRITE_SIZE", 900000.0, 0, 1], [2, 512, MRF, "WRITE_SIZE", 900000.0, 0, 1],
             [3, 512, PRE, "WRITE_SIZE", 50000.0, 0, 1]] + [[10 + i, 1, COPY, "WRITE_SIZE", gib_kib, 0, 1] for i in range(3)]
    _write(str(tmp_path / "f" / "r1_counter_collection.csv"), fetch)
    _write(str(tmp_path / "w" / "r1_counter_collection.csv"), write)
    rec = json.loads(_run("pmc_traffic.py", str(tmp_path / "f"), str(tmp_path / "w"), "2"))
    assert rec["mrf_launches_fetch_pass"] == 2 and rec["mrf_launches_write_pass"] == 2
    assert rec["steps_in_pass"] == 2 and rec["launches_per_step"] == 1      # what bench.py compares with its own count
    assert len(rec["launch_config_digest"]) == 16 and rec["launch_config"]["batch_per_gpu"] == 32
    assert rec["nominal"]["bytes_per_launch"] == (2 * 500000 + 900000) * 1024
    assert abs(rec["calibrated"]["read_factor"] - 2.0) < 1e-6 and abs(rec["calibrated"]["write_factor"] - 1.0) < 1e-6


def test_bench_reports_pmc_traffic_only_for_the_same_launch_configuration(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a committed PMC record: it is reported only when the record was taken on
    the same kernel sources + workload shape + fusion policy (launch_config_digest) AND counted the number of MRF
    launches per step this run issued; a change of PAIR_POLICY / fuse_pairs / batch / frames voids it."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from openvoice_amd.engine import PAIR_POLICY
    from openvoice_amd.hostinfo import launch_config_digest
    rec = {"launch_config_digest": launch_config_digest(32, 861, True, PAIR_POLICY), "launches_per_step": 63,
           "nominal": {"bytes_per_launch": 123}, "calibrated": {"bytes_per_launch": 456}}
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(f))
    assert bench.pmc_traffic(32, 861, True, PAIR_POLICY, 63)[0] == 456
    for args in ((16, 861, True, PAIR_POLICY, 63), (32, 860, True, PAIR_POLICY, 63), (32, 861, False, PAIR_POLICY, 63),
                 (32, 861, True, set(list(PAIR_POLICY)[1:]), 63), (32, 861, True, PAIR_POLICY, 72),
                 (32, 861, True, PAIR_POLICY, 63, 3)):
        value, why = bench.pmc_traffic(*args)
        assert value is None and "this run" in why, args
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILE", str(tmp_path / "missing.json"))
    assert bench.pmc_traffic(32, 861, True, PAIR_POLICY, 63) == (None, "no PMC record committed")


def test_split_conv_traffic_per_instance_and_its_gate_in_bench(tmp_path, monkeypatch):
    """tools/pmc_traffic_split.py: per kernel instance, FETCH_SIZE x the factor measured on the pass's own 1 GiB copies and
    WRITE_SIZE against the 6-bytes-per-element streams (residual form: two tensors read); bench.split_traffic_record hands
    the record out only for the split kernel sources, shape and product count it was measured on."""
    gib_kib = float(1 << 20)
    s128 = "void ovks3::conv1d_split3_kernel<11, 1, 128, 128, true, 6>(ov_conv1d_split3_params)"
    s256 = "void ovks3::conv1d_split3_kernel<3, 5, 256, 128, false, 6>(ov_conv1d_split3_params)"
    tensor128 = 6 * 2 * 10 * 64 * 128            # batch 2, 10 frames, stage 1: 64 columns per frame, C = 128
    tensor256 = 6 * 2 * 10 * 8 * 256
    fetch = [[1, 256, s128, "FETCH_SIZE", 1.5 * 2 * tensor128 / 2 / 1024, 0, 1], [2, 256, s256, "FETCH_SIZE", 3.0 * tensor256 / 2 / 1024, 0, 1]]
    write = [[1, 256, s128, "WRITE_SIZE", tensor128 / 1024, 0, 1], [2, 256, s256, "WRITE_SIZE", tensor256 / 1024, 0, 1]]
    fetch += [[10 + i, 1, COPY, "FETCH_SIZE", gib_kib / 2, 0, 1] for i in range(3)]
    write += [[10 + i, 1, COPY, "WRITE_SIZE", gib_kib, 0, 1] for i in range(3)]
    _write(str(tmp_path / "f" / "r1_counter_collection.csv"), fetch)
    _write(str(tmp_path / "w" / "r1_counter_collection.csv"), write)
    rec = json.loads(_run("pmc_traffic_split.py", str(tmp_path / "f"), str(tmp_path / "w"), "1", "2", "10"))
    by = {(i["C"], i["residual"]): i for i in rec["instances"]}
    assert by[(128, True)]["read_over_algorithmic"] == 1.5 and by[(256, False)]["read_over_algorithmic"] == 3.0
    assert all(i["write_over_algorithmic"] == 1.0 and i["products"] == 6 for i in rec["instances"])
    assert rec["launches_per_step"] == 2 and rec["algorithmic_bytes_per_step"] == 3 * tensor128 + 2 * tensor256
    assert rec["bytes_per_step"] == round(1.5 * 2 * tensor128 + tensor128 + 3.0 * tensor256 + tensor256)
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    os.makedirs(tmp_path / "profiles")
    (tmp_path / "profiles" / "split3_traffic_latest.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    assert bench.split_traffic_record(2, 10, 6)[0]["bytes_per_step"] == rec["bytes_per_step"]
    assert bench.split_traffic_record(2, 10, 3)[0] is None and bench.split_traffic_record(32, 10, 6)[0] is None
    rec["split3_source_digest"] = "0" * 16
    (tmp_path / "profiles" / "split3_traffic_latest.json").write_text(json.dumps(rec))
    assert bench.split_traffic_record(2, 10, 6)[0] is None
    # and the committed record belongs to the committed kernels
    from openvoice_amd.hostinfo import split3_source_digest
    with open(os.path.join(ROOT, "profiles", "split3_traffic_latest.json")) as fh:
        assert json.load(fh)["split3_source_digest"] == split3_source_digest()


def test_committed_counter_records_belong_to_the_committed_sources():
    """bench.py reports a PMC figure only when the committed record's digest equals the running tree's; a record gone stale
    (a kernel source, the C ABI header or engine.py edited without re-running the PMC passes: scripts/gpu_r5_s27.sh /
    gpu_r5_s20.sh / profile_bf16.sh) would silently turn ``roofline.traffic`` into null on the contract line -- fail here
    instead."""
    sys.path.insert(0, ROOT)
    from openvoice_amd.engine import PAIR_POLICY
    from openvoice_amd.hostinfo import bf16_source_digest, launch_config_digest, split3_source_digest
    load = lambda name: json.load(open(os.path.join(ROOT, "profiles", name)))
    assert load("pmc_traffic_latest.json")["launch_config_digest"] == launch_config_digest(32, 861, True, PAIR_POLICY)
    assert load("split3_traffic_latest.json")["split3_source_digest"] == split3_source_digest()
    assert load("bf16_counters_latest.json")["bf16_source_digest"] == bf16_source_digest()
