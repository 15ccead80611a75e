"""bench.py's N > 1 control flow, rehearsed on CPU with gloo: the exact launch line
the driver uses (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...), the
rendezvous, the per-step broadcast of rank 0's speaker embeddings, the barrier / synchronise bracketing, the
max-over-ranks all-reduce of the elapsed time and the single JSON line on rank 0 -- so that the first RCCL run is not
also the first execution of that code.  Only the conversion itself is replaced by a stand-in (--dry-run)."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _lines(stdout):
    return [json.loads(line) for line in stdout.splitlines() if line.startswith("{")]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 4])
def test_bench_multi_rank_control_flow_on_gloo(n):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-run"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=REPO)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = _lines(res.stdout)
    assert len(lines) == 1, f"exactly one JSON line, from rank 0: {res.stdout!r}"
    rec = lines[0]
    assert rec["n_gpus"] == n and rec["steps"] == 3 and rec["warmup"] == 1 and rec["dry_run"] is True
    assert rec["broadcast_consistent"] is True
    assert rec["config"]["global_batch"] == 32 * n and rec["scaling"] == "weak"
    assert rec["ms_per_step"] >= 0
    pr = rec["per_rank_ms_per_step"]         # every rank's own time: a straggler shows up as max >> min
    assert len(pr["ranks"]) == n and pr["min"] <= pr["max"] <= rec["ms_per_step"] + 1e-3
    # the N > 1 line is COMPLETE: roofline-shaped object, the CPU baseline (timed on rank 0 after the process group is
    # gone), every rank's device gathered over the group
    assert set(rec["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    cb = rec["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb and "unit" in cb
    assert rec["distributed"]["process_group"] is True and len(rec["distributed"]["devices"]) == n
    # the unattended N-GPU run diagnoses itself -- every rank's own MRF time and resident bytes come
    # back over the group, and the collective library's version + the environment switches are on the line
    per = rec["distributed"]["per_rank"]
    assert [r["rank"] for r in per] == list(range(n)) and all({"mrf_ms", "hbm_resident_bytes", "device"} <= set(r) for r in per)
    assert {"rccl_version", "env", "torch"} <= set(rec["distributed"])
    assert {"HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_DEBUG_SUBSYS"} <= set(rec["distributed"]["env"])


def test_bench_single_rank_dry_run_needs_no_process_group():
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, cwd=REPO)
    assert res.returncode == 0, res.stderr[-2000:]
    (rec,) = _lines(res.stdout)
    assert rec["n_gpus"] == 1 and rec["dry_run"] is True


def test_force_dist_runs_the_collectives_at_one_rank():
    """--force-dist: process group, broadcast, barriers and the timing all-gather at world size 1, without
    torch.distributed.run (the bench makes its own 1-rank rendezvous); on the GPU box the same flag runs RCCL
    (tests/test_gpu_dist.py)."""
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", "--force-dist", "--gpus", "1",
                          "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=120, cwd=REPO)
    assert res.returncode == 0, res.stderr[-2000:]
    (rec,) = _lines(res.stdout)
    assert rec["n_gpus"] == 1 and rec["broadcast_consistent"] is True and len(rec["per_rank_ms_per_step"]["ranks"]) == 1


def test_zero_steps_is_refused():
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", "--steps", "0"],
                         capture_output=True, text=True, timeout=120, cwd=REPO)
    assert res.returncode != 0 and "--steps" in res.stderr


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", "--gpus", "4"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=REPO)
    assert res.returncode != 0 and "nproc-per-node" in res.stderr


@pytest.mark.timeout(300)
def test_plain_python_launch_with_gpus_2_starts_its_own_ranks():
    """the driver starts the 1-GPU line as ``python3 bench.py --gpus 1 ...``; started the same way
    with ``--gpus 2`` (no torch.distributed.run, no WORLD_SIZE / RANK / MASTER_* in the environment) bench.py must
    re-execute itself under torch.distributed.run with 2 ranks instead of failing on the launch convention, and still
    print exactly ONE complete line from rank 0."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                        "GROUP_RANK", "TORCHELASTIC_RUN_ID")}
    env["OMP_NUM_THREADS"] = "1"
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--dry-run", "--gpus", "2", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=280, env=env, cwd=REPO)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "re-executing under torch.distributed.run" in res.stderr
    (rec,) = _lines(res.stdout)
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["broadcast_consistent"] is True
    d = rec["distributed"]
    assert d["process_group"] is True and d["backend"] == "gloo" and "rccl_ranks" in d and len(d["devices"]) == 2
    assert len(rec["per_rank_ms_per_step"]["ranks"]) == 2
    cb = rec["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("port", "reference") and "batch32" in cb


def test_more_ranks_than_devices_is_one_json_error_line_before_any_launch():
    """``--gpus 8`` on a node with fewer devices (here: none) prints ONE JSON line with "error" and
    exits non-zero BEFORE re-executing under torch.distributed.run -- no rendezvous, no hang."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, env=env, cwd=REPO)
    assert res.returncode == 4, (res.returncode, res.stderr[-500:])
    assert "re-executing" not in res.stderr
    (rec,) = _lines(res.stdout)
    assert "device_count() = 0" in rec["error"] and rec["n_gpus"] == 8 and rec["value"] is None
