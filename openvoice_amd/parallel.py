"""Batch-sharded data parallelism for the converter: one process per GPU, utterances split across
ranks, weights replicated, and ONE collective per batch -- a broadcast of the packed source/target
speaker embeddings (2 x gin fp32 = 2 KiB) from rank 0 over RCCL/xGMI (``torch.distributed`` backend
"nccl" on ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all
(SURVEY.md section 2); utterances are independent, so there is no data-path collective beyond
this broadcast.
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


def broadcast_speaker_embeddings(src_se, tgt_se, gin, device, root=0, group=None):
    """Rank ``root`` supplies (src_se, tgt_se); every rank returns them as [1,gin,1] tensors on
    ``device``.  Non-root ranks may pass ``None``.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return src_se.to(device), tgt_se.to(device)
    if dist.get_rank(group) == root:
        packed = pack_speaker_embeddings(src_se.to(device, torch.float32), tgt_se.to(device, torch.float32))
    else:
        packed = torch.empty(2, gin, dtype=torch.float32, device=device)
    dist.broadcast(packed, src=root, group=group)
    return unpack_speaker_embeddings(packed)


def gather_waveforms(local_wave, group=None):
    """Optional: all-gather equally-sized [B_local, 1, L] outputs into [B_total, 1, L]."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_wave
    out = [torch.empty_like(local_wave) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, local_wave.contiguous(), group=group)
    return torch.cat(out, 0)
