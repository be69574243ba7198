#!/usr/bin/env python3
"""Host-side JPEG entropy front end: pixels per second of ffhip_jpeg_entropy_batch on
the fixture file, by thread count, next to the reference's own whole-file decode when oracle/_ref is
present (build container only).  CPU only."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ffpic_amd import ops

data = open(os.path.join(ROOT, "tests", "golden", "file_q85_420.jpg"), "rb").read()
g, w, h = ops.jpeg_probe(data)
n = 256
files = [data] * n
out = {"file": "file_q85_420.jpg", "coded": [g.width, g.height], "bytes": len(data), "copies": n}
import ctypes as C
from ffpic_amd import capi
L = capi.lib()
bufs = [np.frombuffer(data, np.uint8)] * n
ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
lens = (C.c_size_t * n)(*[b.size for b in bufs])
cy = np.zeros(n * g.y_blocks * 64, np.int16); cu = np.zeros(n * g.c_blocks * 64, np.int16); cv = np.zeros(n * g.c_blocks * 64, np.int16)
quant = np.zeros((n, 4, 64), np.uint16); status = (C.c_int * n)()      # outputs allocated and touched once: no page faults in the timing
for th in (1, 2, 4, 8, 16):
    if th > (os.cpu_count() or 1):
        break
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        capi.check(L.ffhip_jpeg_entropy_batch(ptrs, lens, n, th, C.byref(g), cy.ctypes.data, cu.ctypes.data, cv.ctypes.data, quant.ctypes.data, status))
        best = min(best, time.perf_counter() - t0)
    out[f"threads_{th}"] = {"Mpx/s": round(n * g.width * g.height / best / 1e6, 1), "MB/s": round(n * len(data) / best / 1e6, 1)}
print(json.dumps(out, indent=1))
