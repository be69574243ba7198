"""Data-parallel sharding of a batch of independent images over the GPUs of one node: the Python face of
ffhip_shard.hip (include/ffpic_hip.h, "batches over the GPUs of one node").

The hot path has no exchange step: a batch is split into contiguous image ranges, one per rank, and the
only collective is the batch close -- an all-gather of one 32-byte {rank, status, first, count, checksum} record per
rank that doubles as the batch barrier.  Range arithmetic, the record layout and the "does this tile the batch"
verdict live in C (ffhip_shard_range, ffhip_batch_close, ffhip_batch_complete); the transport is RCCL called from C
(ncclAllGather on the rank's stream) on a multi-GPU node, and torch.distributed (gloo) only where RCCL cannot run: the
CPU tests and the one-GPU rehearsal of bench.py.
"""
import ctypes as C
import os

import numpy as np

from . import capi


def shard_range(n_images, rank, world_size):
    """Contiguous range [first, last) of the images rank owns; sizes differ by at most 1 (ffhip_shard_range)."""
    first, count = C.c_longlong(), C.c_longlong()
    if capi.lib().ffhip_shard_range(n_images, rank, world_size, C.byref(first), C.byref(count)) != 0:
        raise ValueError("bad shard arguments")
    return first.value, first.value + count.value


def batch_complete(records, n_images):
    """True when the records (ctypes array of capi.BatchRecord, one per rank, in rank order) tile [0, n_images)
    exactly and every status is 0 (ffhip_batch_complete)."""
    return bool(capi.lib().ffhip_batch_complete(records, len(records), n_images))


class Batch:
    """One rank's handle on the batch close.  transport: "none" (one GPU), "rccl" (ffhip_batch_close over a
    communicator made by ffhip_comm_init_rank) or "torch" (the record travels through torch.distributed)."""

    def __init__(self, rank=0, world=1, device=None, transport=None):
        self.rank, self.world, self.device = rank, world, device
        self.comm = None
        self.transport = "none"
        if world == 1:
            return
        import torch
        import torch.distributed as dist
        assert dist.is_initialized()
        want = transport or os.environ.get("FFHIP_BATCH_CLOSE") or ("rccl" if dist.get_backend() == "nccl" else "torch")
        self.transport = "torch"
        if want == "rccl":
            L = capi.lib()
            # 128 bytes of id + one byte "rank 0 got an id": ncclCommInitRank is collective, so a rank must only enter it
            # when every rank will -- an id rank 0 failed to make would leave the others blocked in the bootstrap
            ident = torch.zeros(129, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_uint8 * 128)()
                ok = int(L.ffhip_comm_unique_id(buf) == 0)
                ident = torch.from_numpy(np.concatenate([np.frombuffer(buf, dtype=np.uint8), np.array([ok], np.uint8)]))
            ident = ident.to(device) if device is not None else ident
            dist.broadcast(ident, 0)
            host = ident.cpu().numpy()
            if int(host[128]) == 1:
                self.comm = L.ffhip_comm_init_rank(host[:128].tobytes(), rank, world)
            # every rank must take the same road: RCCL from C only if every rank got its communicator
            flag = torch.tensor([1 if self.comm else 0], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                self.transport = "rccl"
            elif self.comm:
                L.ffhip_comm_destroy(self.comm)
                self.comm = None

    def close(self, first, count, status=0, checksum=0, stream=None):
        """Closes the batch behind what `stream` holds; returns the ctypes array of `world` records in rank order."""
        L = capi.lib()
        recs = (capi.BatchRecord * self.world)()
        if self.transport in ("none", "rccl"):
            capi.check(L.ffhip_batch_close(self.comm, self.rank, self.world, first, count, status, checksum, recs, stream),
                       "ffhip_batch_close")
            return recs
        import torch
        import torch.distributed as dist
        if stream is not None or (self.device is not None and L.ffhip_device_count() > 0):
            capi.check(L.ffhip_stream_sync(stream), "ffhip_stream_sync")
        mine = capi.BatchRecord(self.rank, status, first, count, checksum)
        t = torch.from_numpy(np.frombuffer(bytes(mine), dtype=np.uint8).copy())
        if dist.get_backend() == "nccl":
            t = t.to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        for r, o in enumerate(out):
            C.memmove(C.byref(recs[r]), o.cpu().numpy().tobytes(), C.sizeof(capi.BatchRecord))
        return recs

    def destroy(self):
        if self.comm:
            capi.lib().ffhip_comm_destroy(self.comm)
            self.comm = None
