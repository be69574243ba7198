#!/usr/bin/env python3
"""The fused JPEG kernels and their arithmetic-free twins (ffhip_jpeg_pattern_calibrate), every layout into the SAME output buffer (256 x 3840x2176 BGRA, held;
inputs of the layouts' own sizes), a few allocations in turn: which part of a layout's rate is its access pattern's, which the placement's."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from ffpic_amd import capi, ops, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
n, W, H = 256, 3840, 2176
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
layouts = {"420": (3, 2, 2), "411": (3, 4, 1), "114": (3, 1, 4), "444": (3, 1, 1), "422": (3, 2, 1), "440": (3, 1, 2), "grey": (1, 1, 1)}
ins = {}
for name, (nc, h, v) in layouts.items():
    cols, rows = W // (8 * h), H // (8 * v)
    g = capi.jpeg_geom(cols, rows, nc, h, v, (0, 1, 1))
    ty = torch.randint(-30, 31, (n * cols * rows * h * v * 64,), device=dev, dtype=torch.int16)
    tu = torch.randint(-30, 31, (n * cols * rows * 64,), device=dev, dtype=torch.int16) if nc == 3 else None
    tv = torch.randint(-30, 31, (n * cols * rows * 64,), device=dev, dtype=torch.int16) if nc == 3 else None
    ins[name] = (g, ty, tu, tv, 4 + 2 * (1 + (2.0 / (h * v) if nc == 3 else 0)))
def timed(name, out, pattern, reps=6):
    g, ty, tu, tv, bpp = ins[name]
    up, vp = (tu.data_ptr(), tv.data_ptr()) if tu is not None else (None, None)
    def step():
        if pattern: capi.check(L.ffhip_jpeg_pattern_calibrate(C.byref(g), n, ty.data_ptr(), up, vp, q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H, st))
        else: ops.jpeg_recon_batch(g, n, ty.data_ptr(), up, vp, q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H, None, 0, st)
    for _ in range(2): step()
    L.ffhip_event_record(e0, st)
    for _ in range(reps): step()
    L.ffhip_event_record(e1, st); capi.check(L.ffhip_stream_sync(st))
    return round(bpp * n * W * H / (L.ffhip_event_elapsed_ms(e0, e1) / reps) / 1e9, 3)
held = []
for a in range(int(os.environ.get("ALLOCS", "4"))):
    out = torch.empty(n * W * H * 4, dtype=torch.uint8, device=dev)
    held.append(out)
    row = {"allocation": a}
    for name in layouts:
        row[name] = {"kernel_TB/s": timed(name, out, False), "pattern_TB/s": timed(name, out, True)}
        if os.environ.get("STRIPS_AB") and name not in ("420", "411", "114"):      # one and two strips' worth per wave on the same buffer (FFHIP_JPEG_STRIPS)
            for sv in ("1", "2"):
                capi.setenv("FFHIP_JPEG_STRIPS", sv)
                row[name][f"strips{sv}"] = (timed(name, out, False), timed(name, out, True))
            capi.setenv("FFHIP_JPEG_STRIPS", None)
    print(json.dumps(row), flush=True)
