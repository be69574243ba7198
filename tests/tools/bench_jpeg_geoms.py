#!/usr/bin/env python3
"""Throughput of ffhip_jpeg_recon_batch for the sampling layouts other than 4:2:0 (generic path) next to
4:2:0 (fused kernel), 64 x 3840x2160 each, inputs resident in HBM, HIP events on the launch stream.
Algorithmic bytes per pixel: 2 B per coefficient sample + 4 B BGRA (4:4:4 10, 4:2:2 / 4:4:0 8, 4:2:0 7, grey 6)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])   # A/B runs of two builds in one gpurun call

dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
n = int(os.environ.get("FFHIP_BENCH_IMAGES", "64"))
out = {}
for name, (nc, h, v) in {"420": (3, 2, 2), "444": (3, 1, 1), "422": (3, 2, 1), "440": (3, 1, 2), "411": (3, 4, 1), "114": (3, 1, 4), "grey": (1, 1, 1)}.items():
    W, H = 3840, 2176
    cols, rows = W // (8 * h), H // (8 * v)
    g = capi.jpeg_geom(cols, rows, nc, h, v, (0, 1, 1))
    by = cols * rows * h * v
    ty = torch.randint(-30, 31, (n * by, 64), device=dev).to(torch.int16)
    tu = torch.randint(-30, 31, (n * cols * rows, 64), device=dev).to(torch.int16) if nc == 3 else None
    tv = tu.clone() if nc == 3 else None
    q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
    o = torch.empty(n * W * H * 4, dtype=torch.uint8, device=dev)
    wsb = L.ffhip_jpeg_workspace_bytes(C_byref := __import__("ctypes").byref(g), n)
    ws = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device=dev)
    def run():
        ops.jpeg_recon_batch(g, n, ty.data_ptr(), tu.data_ptr() if nc == 3 else None, tv.data_ptr() if nc == 3 else None, q.data_ptr(), 0,
                             o.data_ptr(), W * 4, W * 4 * H, ws.data_ptr(), int(wsb), st)
    for _ in range(3): run()
    L.ffhip_event_record(e0, st)
    for _ in range(10): run()
    L.ffhip_event_record(e1, st)
    ms = L.ffhip_event_elapsed_ms(e0, e1) / 10
    px = n * W * H
    bpp = 4 + 2 * (1 + (2.0 / (h * v) if nc == 3 else 0))
    out[name] = {"ms": round(ms, 3), "Gpx/s": round(px / ms / 1e6, 1), "GB/s": round(px * bpp / ms / 1e6, 1), "workspace_MB": round(int(wsb) / 1e6, 1)}
print(json.dumps(out, indent=1))
