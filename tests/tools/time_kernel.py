#!/usr/bin/env python3
"""Time ffhip_jpeg_recon_batch on a resident batch (HIP events on the launch stream) and check the
first image against the CPU oracle.  Diagnostic helper for kernel-variant A/B runs:
    FFHIP_JPEG_VARIANT=21 python tests/tools/time_kernel.py --workload c3 --steps 10 --rounds 3"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ffpic_amd import capi, ops
import bench, oracle_lib as O

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3"); ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--images", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = capi.require_device(0)
cols, rows, n, _ = bench.WORKLOADS[a.workload]
n = a.images or n
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
t_y, t_u, t_v, q = bench.gen_batch_on_device(dev, cols, rows, n, 0)
t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
out = torch.empty(n * W * 4 * H, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def step():
    ops.jpeg_recon_batch(geom, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0, out.data_ptr(), W*4, W*4*H, None, 0, st)
step(); torch.cuda.synchronize()
m = cols * rows
exp = O.oracle_jpeg_recon(O.make_geom(cols, rows), t_y[:m*256].cpu().numpy(), t_u[:m*64].cpu().numpy(), t_v[:m*64].cpu().numpy(), q)[0]
ok = np.array_equal(out[:W*4*H].cpu().numpy().reshape(H, W, 4), exp)
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
res = []
for r in range(a.rounds):
    for _ in range(3): step()
    L.ffhip_event_record(e0, st)
    for _ in range(a.steps): step()
    L.ffhip_event_record(e1, st)
    res.append(L.ffhip_event_elapsed_ms(e0, e1) / a.steps)
best = min(res)
print(f"variant={os.environ.get('FFHIP_JPEG_VARIANT','default')} parity={ok} ms={['%.4f'%x for x in res]} best={best:.4f} "
      f"GB/s={7.0*n*H*W/(best*1e-3)/1e9:.0f} Gpx/s={n*H*W/(best*1e-3)/1e9:.1f}")
