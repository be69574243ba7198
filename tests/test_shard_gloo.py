"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py shards a batch of
independent images into contiguous per-rank ranges with no data-path collective and
closes the batch with one all-gather of status records.  Each rank reconstructs its
shard (with the CPU checker standing in for the device here) and the union must equal
the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from ffpic_amd import shard, synth


def test_shard_range_tiles_everything():
    for n in (0, 1, 7, 8, 255, 256, 1024):
        for world in (1, 2, 3, 8):
            pos = 0
            for r in range(world):
                a, b = shard.shard_range(n, r, world)
                assert a == pos and b >= a
                pos = b
            assert pos == n
            sizes = [shard.shard_range(n, r, world)[1] - shard.shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(4, 2, 2)


def test_shard_range_refuses_nonsense_through_the_c_abi():
    import ctypes as C
    from ffpic_amd import capi
    L = capi.lib()
    a, b = C.c_longlong(), C.c_longlong()
    assert L.ffhip_shard_range(10, 0, 0, C.byref(a), C.byref(b)) == capi.FFHIP_EINVAL
    assert L.ffhip_shard_range(-1, 0, 1, C.byref(a), C.byref(b)) == capi.FFHIP_EINVAL
    assert L.ffhip_shard_range(10, 1, 1, C.byref(a), C.byref(b)) == capi.FFHIP_EINVAL
    assert L.ffhip_shard_range(10, 0, 1, None, C.byref(b)) == capi.FFHIP_EINVAL
    assert L.ffhip_shard_range(256, 7, 8, C.byref(a), C.byref(b)) == 0 and (a.value, b.value) == (224, 32)   # BASELINE config 3


def test_one_gpu_close_and_record_logic():
    """ffhip_batch_close without a communicator (the one-GPU case; needs no device) and ffhip_batch_complete on
    hand-made record sets: gaps, overlaps, a failing rank, a record in the wrong slot, empty ranges"""
    from ffpic_amd import capi
    b = shard.Batch()
    recs = b.close(0, 5, 0, checksum=0xDEADBEEF)
    assert len(recs) == 1 and (recs[0].rank, recs[0].first, recs[0].count, recs[0].status, recs[0].checksum) == (0, 0, 5, 0, 0xDEADBEEF)
    assert shard.batch_complete(recs, 5) and not shard.batch_complete(recs, 6)

    def mk(*items):
        arr = (capi.BatchRecord * len(items))()
        for i, (rank, status, first, count) in enumerate(items):
            arr[i] = capi.BatchRecord(rank, status, first, count, 0)
        return arr
    assert shard.batch_complete(mk((0, 0, 0, 3), (1, 0, 3, 2)), 5)
    assert shard.batch_complete(mk((0, 0, 0, 1), (1, 0, 1, 0), (2, 0, 1, 0)), 1)          # more ranks than images
    assert shard.batch_complete(mk((0, 0, 0, 0)), 0)
    assert not shard.batch_complete(mk((0, 0, 0, 3), (1, 0, 4, 1)), 5)                      # gap
    assert not shard.batch_complete(mk((0, 0, 0, 3), (1, 0, 2, 3)), 5)                      # overlap
    assert not shard.batch_complete(mk((0, 0, 0, 3), (1, -5, 3, 2)), 5)                     # a failing rank
    assert not shard.batch_complete(mk((1, 0, 3, 2), (0, 0, 0, 3)), 5)                      # records not in rank order
    assert not shard.batch_complete(mk((0, 0, 0, 3), (1, 0, 3, 3)), 5)                      # beyond the batch
    assert not shard.batch_complete(mk((0, 0, 0, 3), (1, 0, 3, -1)), 2)
    import ctypes as C
    assert capi.lib().ffhip_batch_close(None, 0, 2, 0, 1, 0, 0, recs, None) == capi.FFHIP_EINVAL   # world 2 needs a communicator
    assert C.sizeof(capi.BatchRecord) == 32


def _worker(rank, world, port, n_images, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cols, rows = 4, 2
    first, last = shard.shard_range(n_images, rank, world)
    cy, cu, cv = synth.coef_batch(last - first, cols, rows, first=first)
    out = O.oracle_jpeg_recon(O.make_geom(cols, rows), cy, cu, cv, synth.quant_tables(), n_images=last - first)
    np.save(os.path.join(tmpdir, f"part{rank}.npy"), out)
    batch = shard.Batch(rank, world)                                   # gloo: the record travels through torch.distributed
    assert batch.transport == "torch"
    recs = batch.close(first, last - first, 0, checksum=int(out.astype(np.uint64).sum()))
    assert len(recs) == world and [r.rank for r in recs] == list(range(world))
    assert shard.batch_complete(recs, n_images)
    assert recs[rank].checksum == int(out.astype(np.uint64).sum())
    bad = batch.close(first, last - first, 0 if rank else -5)          # one failing rank fails the batch
    assert not shard.batch_complete(bad, n_images)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n_images, world = 5, 2
    mp.spawn(_worker, args=(world, port, n_images, str(tmp_path)), nprocs=world, join=True)
    cols, rows = 4, 2
    cy, cu, cv = synth.coef_batch(n_images, cols, rows)
    whole = O.oracle_jpeg_recon(O.make_geom(cols, rows), cy, cu, cv, synth.quant_tables(), n_images=n_images)
    parts = np.concatenate([np.load(tmp_path / f"part{r}.npy") for r in range(world)])
    assert np.array_equal(parts, whole)
