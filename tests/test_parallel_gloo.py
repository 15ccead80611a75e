"""Multi-process (world_size 2, gloo, CPU) checks of the data-parallel layer: utterance sharding and
the one collective on the path, the broadcast of the packed src/tgt speaker embeddings
(openvoice_amd/parallel.py).  On the GPU box the same code runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openvoice_amd.parallel import (broadcast_speaker_embeddings, gather_waveforms, pack_speaker_embeddings,
                                    shard_range, unpack_speaker_embeddings)


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_round_trip():
    src, tgt = torch.randn(1, 256, 1), torch.randn(1, 256, 1)
    a, b = unpack_speaker_embeddings(pack_speaker_embeddings(src, tgt))
    assert torch.equal(a, src) and torch.equal(b, tgt)


def test_broadcast_is_identity_without_process_group():
    src, tgt = torch.randn(1, 256, 1), torch.randn(1, 256, 1)
    a, b = broadcast_speaker_embeddings(src, tgt, 256, "cpu")
    assert torch.equal(a, src) and torch.equal(b, tgt)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gen = torch.Generator().manual_seed(11)
        src, tgt = torch.randn(1, 256, 1, generator=gen), torch.randn(1, 256, 1, generator=gen)
        # only rank 0 owns the embeddings; everyone must end up with identical copies
        a, b = broadcast_speaker_embeddings(src if rank == 0 else None, tgt if rank == 0 else None, 256, "cpu")
        assert a.shape == (1, 256, 1) and torch.equal(a, src) and torch.equal(b, tgt)
        # shard 5 utterances, "convert" them (stand-in: scale by the embedding mean), gather, compare
        total = 6
        utts = torch.arange(total * 8, dtype=torch.float32).reshape(total, 1, 8)
        lo, hi = shard_range(total, rank, world)
        local = utts[lo:hi] * a.mean()
        full = gather_waveforms(local)
        assert torch.allclose(full, utts * src.mean())
        torch.save(torch.tensor([lo, hi]), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_broadcast_and_gather_world_size_2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    spans = [torch.load(tmp_path / f"rank{r}.pt").tolist() for r in range(world)]
    assert spans == [[0, 3], [3, 6]]
