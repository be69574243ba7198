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


def test_gather_status_without_process_group():
    rec = shard.gather_status(0, 5, 0)
    assert rec.shape == (1, 4) and shard.batch_complete(rec, 5) and not shard.batch_complete(rec, 6)


def _worker(rank, world, port, n_images, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cols, rows = 4, 2
    first, last = shard.shard_range(n_images, rank, world)
    cy, cu, cv = synth.coef_batch(last - first, cols, rows, first=first)
    out = O.oracle_jpeg_recon(O.make_geom(cols, rows), cy, cu, cv, synth.quant_tables(), n_images=last - first)
    np.save(os.path.join(tmpdir, f"part{rank}.npy"), out)
    recs = shard.gather_status(first, last - first, 0)
    assert recs.shape == (world, 4)
    assert shard.batch_complete(recs, n_images)
    bad = shard.gather_status(first, last - first, 0 if rank else -5)   # one failing rank fails the batch
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
