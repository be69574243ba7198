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
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["scaling"] == "strong"
    assert rec["config"]["batch_complete"] is True and rec["config"]["parity_vs_oracle_first_and_last_image"] is True
    assert rec["config"]["images_total"] == 5 and rec["roofline"]["bound"] == "hbm" and rec["roofline"]["frac"] > 0


@pytest.mark.parametrize("scaling,total", [("strong", 7), ("weak", 6)])
def test_two_rank_rehearsal(scaling, total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, FFHIP_BENCH_REHEARSE="1")
    images = 7 if scaling == "strong" else 3
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--images", str(images), "--scaling", scaling, "--no-cpu"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    rec = _line(out)
    assert rec["n_gpus"] == 2 and rec["scaling"] == scaling and "rehearsal" in rec["config"]
    assert rec["config"]["images_total"] == total and rec["config"]["batch_complete"] is True
    assert rec["config"]["parity_vs_oracle_first_and_last_image"] is True and "extra" not in rec
