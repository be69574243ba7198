#!/usr/bin/env python3
"""N (default 256) copies of one 4K baseline JPEG file WITHOUT restart markers through ffhip_jpeg_entropy_batch_gpu, three times; prints the last
call's phases (ffhip_debug_huff_times).  The file is bench.py's configs.f1 one.  For rocprofv3 --kernel-trace (prof_huff_plain_timeline.sh).
env: N, W, H, QUALITY (85), NOISE (6), STREAM (a stream of its own instead of the null stream), RESTART_ROWS (restart markers every so many MCU rows)"""
import ctypes as C, io, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from PIL import Image
from ffpic_amd import capi

n = int(os.environ.get("N", 256)); W = int(os.environ.get("W", 3840)); H = int(os.environ.get("H", 2160))
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:H, 0:W]
img = np.stack([128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), 128 + 90 * np.cos(xx / 11.0 + yy / 53.0), (xx * 255 / (W - 1) + yy * 255 / (H - 1)) / 2], axis=2)
img = np.clip(img + rng.normal(0, float(os.environ.get("NOISE", 6)), img.shape), 0, 255).astype(np.uint8)
bio = io.BytesIO()
kw = {"restart_marker_rows": int(os.environ["RESTART_ROWS"])} if os.environ.get("RESTART_ROWS") else {}
Image.fromarray(img).save(bio, "JPEG", quality=int(os.environ.get("QUALITY", 85)), subsampling=2, **kw)
data = bio.getvalue()
L = capi.require_device()
buf = np.frombuffer(data, dtype=np.uint8)
vp = C.c_void_p
ptrs = (vp * n)(*([buf.ctypes.data] * n)); lens = (C.c_size_t * n)(*([buf.size] * n)); status = (C.c_int * n)()
g = capi.JpegGeom(); w = C.c_int(); h = C.c_int()
capi.check(L.ffhip_jpeg_probe(buf.ctypes.data, buf.size, C.byref(g), C.byref(w), C.byref(h)))
dev = torch.device("cuda:0")
yb, cb = g.mcu_cols * g.mcu_rows * 4 * 64, g.mcu_cols * g.mcu_rows * 64
d_y = torch.empty(n * yb, dtype=torch.int16, device=dev); d_u = torch.empty(n * cb, dtype=torch.int16, device=dev)
d_v = torch.empty(n * cb, dtype=torch.int16, device=dev); d_q = torch.empty(n * 256, dtype=torch.int16, device=dev)
st = torch.cuda.Stream(device=dev).cuda_stream if os.environ.get("STREAM") else None
best, ph = 1e9, None
for _ in range(3):
    t0 = time.perf_counter()
    capi.check(L.ffhip_jpeg_entropy_batch_gpu(ptrs, lens, n, 16, C.byref(g), d_y.data_ptr(), d_u.data_ptr(), d_v.data_ptr(), d_q.data_ptr(), status, st))
    dt = time.perf_counter() - t0
    if dt < best:
        best = dt
        tt = (C.c_double * 8)(); L.ffhip_debug_huff_times(tt); ph = [round(float(x) / 1e3, 3) for x in tt]
print(json.dumps({"files": n, "file_bytes": len(data), "call_ms": round(best * 1e3, 2), "phases_ms": ph, "device_ms": ph[6]}))
