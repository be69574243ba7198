"""Data-parallel sharding of a batch of independent images over the GPUs of one node.

The hot path has no exchange step: images are independent, so a batch
is split into contiguous image ranges, one per rank, and the only collective is a tiny
all-gather of per-rank {rank, first, count, status} records that doubles as the batch
barrier (RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_images, rank, world_size):
    """Contiguous range [first, last) of the images rank owns; sizes differ by at most 1."""
    if world_size < 1 or not 0 <= rank < world_size or n_images < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_images, world_size)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def gather_status(first, count, status, device=None):
    """All-gather one int64[4] record per rank; returns a [world, 4] tensor on `device`.
    With an uninitialised process group (single GPU) it is a local no-op."""
    rec = torch.tensor([0, first, count, status], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return rec.view(1, 4)
    rec[0] = dist.get_rank()
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out)


def batch_complete(records, n_images):
    """True when the gathered records tile [0, n_images) exactly and every status is 0."""
    recs = sorted((int(r[1]), int(r[2]), int(r[3])) for r in records.cpu())
    pos = 0
    for first, count, status in recs:
        if first != pos or status != 0 or count < 0:
            return False
        pos += count
    return pos == n_images
