"""The multi-GPU leg (BASELINE.json configs[2]) on the ONE GPU a test box has: torch.distributed
with backend "nccl" (= RCCL on ROCm) initialised at world size 1, the real speaker-embedding broadcast from a device
buffer, the barriers and the all-gather of ``bench.py``'s timing -- so that the driver's 8-GPU run is not the first
execution of any of that code.  (World size > 1 is covered on CPU + gloo: tests/test_parallel_gloo.py,
tests/test_bench_dry_run.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_broadcast_gather_and_sharded_api_at_world_size_1(synth_sd, tmp_path):
    """In THIS process (librccl shows up among its loaded libraries): init_process_group("nccl", device_id=...),
    ``broadcast_speaker_embeddings`` from a device buffer, ``gather_waveforms``, and the public sharded entry
    ``ToneColorConverter.convert_batch_sharded`` == ``convert_batch``."""
    import torch.distributed as dist
    from openvoice_amd import api, parallel
    from openvoice_amd.utils import default_converter_hparams
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", device_id=torch.device(DEV))
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        gen = torch.Generator().manual_seed(5)
        src, tgt = 0.3 * torch.randn(1, 256, 1, generator=gen), 0.3 * torch.randn(1, 256, 1, generator=gen)
        a, b = parallel.broadcast_speaker_embeddings(src, tgt, 256, torch.device(DEV))
        dist.barrier()
        torch.cuda.synchronize()
        assert a.is_cuda and torch.equal(a.cpu(), src) and torch.equal(b.cpu(), tgt)
        w = torch.randn(3, 1, 512, device=DEV)
        assert torch.equal(parallel.gather_waveforms(w), w)
        t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=DEV)
        out = [torch.empty_like(t)]
        dist.all_gather(out, t)                       # bench.py's per-rank timing exchange
        assert torch.equal(out[0], t)
        # the public sharded entry point
        hps = default_converter_hparams("v2")
        cfg = {"_version_": "v2", "data": dict(hps.data.items()), "model": dict(hps.model.items())}
        (tmp_path / "config.json").write_text(json.dumps(cfg))
        torch.save({"model": synth_sd}, tmp_path / "checkpoint.pth")
        tcc = api.ToneColorConverter(str(tmp_path / "config.json"), device=DEV, enable_watermark=False)
        tcc.load_ckpt(str(tmp_path / "checkpoint.pth"))
        waves = 0.3 * torch.randn(3, 256 * 20, generator=gen)
        noise = torch.randn(3, 192, 20, generator=gen)
        full = tcc.convert_batch_sharded(waves, src, tgt, tau=0.3, noise=noise)
        plain = tcc.convert_batch(waves, src.to(DEV), tgt.to(DEV), tau=0.3, noise=noise)[0]
        assert full.shape == (3, 1, 256 * 20) and torch.equal(full, plain)
        # ragged list: same call, padded to the longest utterance of the WHOLE batch, lengths returned
        ragged = [waves[0], waves[1][: 256 * 11 + 40], waves[2][: 256 * 7]]
        o_r, n_r = tcc.convert_batch_sharded(ragged, src, tgt, tau=0.0)
        o_p, n_p = tcc.convert_batch(ragged, src.to(DEV), tgt.to(DEV), tau=0.0)
        assert torch.equal(n_r, n_p) and n_r.tolist() == [256 * 20, 256 * 11, 256 * 7] and torch.equal(o_r, o_p)
    finally:
        dist.destroy_process_group()


def _bench(extra, launcher):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("OPENVOICE_AMD_BINDING", None)
    cmd = launcher + [os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                      "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(1800)
def test_bench_under_torch_distributed_run_with_rccl_equals_the_plain_run():
    """The driver's launch line at N = 1 with --force-dist: `python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1 --force-dist` runs the RCCL rendezvous (``device_id=``), the per-step broadcast, the barriers and
    the timing all-gather; its contract line must agree with the plain run's (same box, back to back) and both must
    pass the oracle self-check."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    dist_line = _bench(["--force-dist"], launcher)
    plain = _bench([], [sys.executable])
    print("force-dist:", dist_line["ms_per_step"], "ms   plain:", plain["ms_per_step"], "ms")
    assert dist_line["distributed"]["process_group"] is True and dist_line["distributed"]["backend"] == "nccl"
    assert plain["distributed"]["process_group"] is False
    for line in (dist_line, plain):
        assert line["n_gpus"] == 1 and line["parity"]["ok"] and line["parity"]["max_abs_vs_oracle"] <= 1e-3
        assert line["per_rank_ms_per_step"]["ranks"] and line["roofline"]["frac"] > 0.5
    # one 2 KiB broadcast per 140 ms step: the two lines should agree to the run-to-run spread of one box (~1 %).
    # A correctness suite must not fail on clock / power variance of a shared box: the figure is reported, and only a
    # gross disagreement (a collective serialising the step) fails
    ratio = dist_line["ms_per_step"] / plain["ms_per_step"]
    print(f"forced-dist / plain ms per step = {ratio:.4f}")
    assert 0.8 <= ratio <= 1.25
