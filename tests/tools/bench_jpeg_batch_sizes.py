#!/usr/bin/env python3
"""Headline kernel (4K 4:2:0 grids) against the number of images per launch: per-launch time by HIP events over 20
back-to-back launches.  The strong-scaling bench gives every GPU 32 images at N = 8 (BASELINE config 3)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
cols, rows = 240, 135
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
mcus = cols * rows
N = 256
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
ty = torch.randint(-30, 31, (N * mcus * 4, 64), device=dev, dtype=torch.int16)
tu = torch.randint(-30, 31, (N * mcus, 64), device=dev, dtype=torch.int16)
tv = torch.randint(-30, 31, (N * mcus, 64), device=dev, dtype=torch.int16)
out = torch.empty(N * W * 4 * H, dtype=torch.uint8, device=dev)
res = {}
for n in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    def step():
        ops.jpeg_recon_batch(geom, n, ty.data_ptr(), tu.data_ptr(), tv.data_ptr(), q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H, None, 0, st)
    for _ in range(5): step()
    L.ffhip_event_record(e0, st)
    for _ in range(20): step()
    L.ffhip_event_record(e1, st)
    ms = L.ffhip_event_elapsed_ms(e0, e1) / 20
    res[n] = {"ms": round(ms, 4), "TB/s": round(7 * n * W * H / ms / 1e9, 3), "us_per_image": round(ms * 1e3 / n, 2)}
print(json.dumps(res, indent=1))
