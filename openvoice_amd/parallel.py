"""Batch-sharded data parallelism for the converter: one process per GPU, utterances split across
ranks, weights replicated, and ONE collective per batch -- a broadcast of the packed source/target
speaker embeddings (2 x gin fp32 = 2 KiB) from rank 0 over RCCL/xGMI (``torch.distributed`` backend
"nccl" on ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all
(SURVEY.md section 2); utterances are independent, so there is no data-path collective beyond
this broadcast.  ``convert_sharded`` is the composition the public API uses
(``ToneColorConverter.convert_batch_sharded``): shard -> broadcast -> convert the local shard -> optional all-gather.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous, balanced [start, end) slice of ``n_items`` utterances owned by ``rank``."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def pack_speaker_embeddings(src_se, tgt_se):
    """[1,gin,1] + [1,gin,1] -> one contiguous [2,gin] buffer (a single small message)."""
    return torch.stack([src_se.reshape(-1), tgt_se.reshape(-1)]).contiguous()


def unpack_speaker_embeddings(packed):
    return packed[0].reshape(1, -1, 1), packed[1].reshape(1, -1, 1)


def _initialised():
    return dist.is_available() and dist.is_initialized()


def broadcast_speaker_embeddings(src_se, tgt_se, gin, device, root=0, group=None):
    """Rank ``root`` supplies (src_se, tgt_se); every rank returns them as [1,gin,1] tensors on
    ``device``.  Non-root ranks may pass ``None``.  Whenever a process group is initialised the collective is
    issued -- also at world size 1, where it costs one RCCL launch and exercises exactly the code path of the
    N-GPU run (``bench.py --force-dist``); without a process group it is a device copy."""
    if not _initialised():
        return src_se.to(device), tgt_se.to(device)
    if dist.get_rank(group) == root:
        packed = pack_speaker_embeddings(src_se.to(device, torch.float32), tgt_se.to(device, torch.float32))
    else:
        packed = torch.empty(2, gin, dtype=torch.float32, device=device)
    dist.broadcast(packed, src=root, group=group)
    return unpack_speaker_embeddings(packed)


def gather_waveforms(local_wave, group=None, counts=None):
    """Optional: all-gather the ranks' [B_local, 1, L] outputs into [B_total, 1, L] (rank order = utterance order
    under ``shard_range``).  ``counts`` = utterances per rank when the shards are unequal (the shorter ones are padded
    for the collective and trimmed afterwards); ``L`` must agree across ranks."""
    if not _initialised() or dist.get_world_size(group) == 1:
        return local_wave
    world = dist.get_world_size(group)
    if counts is None:
        counts = [local_wave.shape[0]] * world
    biggest = max(counts)
    send = local_wave.contiguous()
    if send.shape[0] < biggest:
        send = torch.cat([send, send.new_zeros(biggest - send.shape[0], *send.shape[1:])])
    out = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(out, send, group=group)
    return torch.cat([o[:n] for o, n in zip(out, counts)], 0)


def convert_sharded(convert_fn, waveforms, src_se, tgt_se, gin, device, noise=None, gather=True, root=0, group=None,
                    frames=None):
    """Convert a batch of independent utterances across the ranks of the process group.

    Every rank passes the same ``waveforms`` ([N, samples] tensor, or a list of N waveforms) and, optionally, the same
    per-utterance ``noise`` [N, C, T]; rank ``root`` supplies the speaker embeddings (others may pass None).  Rank r
    converts utterances ``shard_range(N, r, world)`` with ``convert_fn(waveforms_shard, src_se, tgt_se, noise_shard)
    -> [n_r, 1, L]`` -- ``ToneColorConverter.convert_batch`` on the GPU path, the CPU oracle in the gloo test -- after
    ONE broadcast of the packed embeddings.  Returns the full [N, 1, L] batch on every rank when ``gather`` (one
    all-gather, outside the hot path), else the local shard and its [start, end).

    ``frames`` (ragged batches): every utterance's frame count.  ``noise`` is [N, C, max(frames)] -- the width of the
    WHOLE batch -- while a shard is converted at its OWN longest length, so the noise handed to ``convert_fn`` is cut
    to ``max(frames[start:end])`` frames: per-utterance noise, the same result however the batch is cut."""
    world = dist.get_world_size(group) if _initialised() else 1
    rank = dist.get_rank(group) if _initialised() else 0
    n = len(waveforms)
    start, end = shard_range(n, rank, world)
    src_se, tgt_se = broadcast_speaker_embeddings(src_se, tgt_se, gin, device, root=root, group=group)
    shard = waveforms[start:end]
    nz = None if noise is None else noise[start:end]
    if nz is not None and frames is not None and end > start:
        nz = nz[:, :, :max(int(f) for f in frames[start:end])]
    local = convert_fn(shard, src_se, tgt_se, nz)
    if not gather:
        return local, (start, end)
    counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    return gather_waveforms(local, group=group, counts=counts)
