"""GPU: bench.py's own control paths at small sizes -- the one-GPU line, and the N = 2 path (sharding through
ffhip_shard_range, barriers, the batch close, rank-0-only legs) rehearsed with both ranks on cuda:0 over gloo, since RCCL
refuses two ranks on one device.  The rehearsal's numbers mean nothing; what is checked is that the batch tiles, every
rank's first and last image are the oracle's and ONE JSON line comes out."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(lines[0])


def test_one_gpu_line_small():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--images", "5", "--no-cpu", "--no-extra"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    rec = _line(out)
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert rec["config"]["batch_complete"] is True and rec["config"]["parity_vs_oracle_first_and_last_image"] is True
    assert rec["config"]["images_total"] == 5 and rec["roofline"]["bound"] == "hbm" and rec["roofline"]["frac"] > 0


@pytest.mark.parametrize("scaling,total,launcher", [("strong", 7, "self"), ("weak", 6, "torchrun")])
def test_two_rank_rehearsal(scaling, total, launcher):
    """launcher "self": `python3 bench.py --gpus 2 ...` with no launcher environment -- the way the driver starts the one-GPU line -- must
    spawn its own two ranks (a child torch.distributed.run; the parent never touches the GPU), relay rank 0's ONE JSON line and exit 0;
    "torchrun": started under torch.distributed.run by the caller, as the contract's N > 1 command line does."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FFHIP_BENCH_REHEARSE"] = "1"
    images = 7 if scaling == "strong" else 3
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--images", str(images), "--scaling", scaling, "--no-cpu"]
    head = [sys.executable] if launcher == "self" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                        "--master-port", str(port)]
    out = subprocess.run(head + tail, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    rec = _line(out)
    assert rec["n_gpus"] == 2 and rec["scaling"] == scaling and "rehearsal" in rec["config"]
    assert rec["config"]["images_total"] == total and rec["config"]["batch_complete"] is True
    assert rec["config"]["parity_vs_oracle_first_and_last_image"] is True and "extra" not in rec


_RCCL_SNIPPET = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
if %r:
    import torch
    torch.zeros(1, device="cuda:0")           # torch's own RCCL is now mapped: the library must reuse it
from ffpic_amd import capi
L = capi.require_device()
ident = (C.c_uint8 * 128)()
capi.check(L.ffhip_comm_unique_id(ident), "unique id")
comm = L.ffhip_comm_init_rank(ident, 0, 1)
assert comm, "ncclCommInitRank failed"
st = L.ffhip_stream_create()
recs = (capi.BatchRecord * 1)()
for k in range(3):
    capi.check(L.ffhip_batch_close(comm, 0, 1, 0, 9, 0, 0xABCDEF0123456789 + k, recs, st), "close")
    assert (recs[0].rank, recs[0].status, recs[0].first, recs[0].count, recs[0].checksum) == (0, 0, 0, 9, 0xABCDEF0123456789 + k)
assert L.ffhip_batch_complete(recs, 1, 9) == 1
assert L.ffhip_batch_close(comm, 0, 2, 0, 9, 0, 0, recs, st) == capi.FFHIP_EINVAL      # not this communicator's world
L.ffhip_comm_destroy(comm)
L.ffhip_stream_destroy(st)
print("rccl-from-c ok")
"""


@pytest.mark.parametrize("with_torch", [False, True])
def test_rccl_batch_close_from_c_single_rank(with_torch):
    """ffhip_comm_unique_id / ffhip_comm_init_rank / ffhip_batch_close with a real RCCL communicator of one rank: the
    run-time binding (a fresh dlopen in a plain process, the copy PyTorch already mapped in a torch process), the
    ncclAllGather on the caller's stream and the record round trip.  More ranks need more GPUs than this box has."""
    out = subprocess.run([sys.executable, "-c", _RCCL_SNIPPET % (ROOT, with_torch)], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0 and "rccl-from-c ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_bgra_checksum_matches_its_definition():
    import numpy as np
    from ffpic_amd import capi, ops
    L = capi.require_device()
    rng = np.random.default_rng(4)
    n, H, W, pitch = 3, 37, 53, 53 * 4 + 16
    stride = pitch * H + 64
    buf = rng.integers(0, 256, size=n * stride, dtype=np.uint8)
    d = ops.DeviceBuffer(buf)
    s = ops.DeviceBuffer(nbytes=8 * n)
    capi.check(L.ffhip_bgra_checksum(d.ptr, pitch, stride, W, H, n, s.ptr, None))
    got = s.to_host((n,), np.uint64)
    for i in range(n):
        img = buf[i * stride:i * stride + pitch * H].reshape(H, pitch)[:, :W * 4].copy().view(np.uint32).reshape(-1).astype(np.uint64)
        want = (img * ((np.arange(img.size, dtype=np.uint64) & np.uint64(0xFFFF)) + np.uint64(1))).sum(dtype=np.uint64)
        assert int(got[i]) == int(want), i
    assert L.ffhip_bgra_checksum(d.ptr, W * 4 - 4, stride, W, H, n, s.ptr, None) == capi.FFHIP_EINVAL
