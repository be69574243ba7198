#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point ffhip_jpeg_recon_batch_host (pageable numpy buffers in,
BGRA out, device buffers allocated and freed inside the call): never the headline number, quoted in DESIGN.md 5."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ffpic_amd import capi, ops, synth

capi.require_device(0)
cols, rows, n = 240, 135, 16
g = capi.jpeg_geom(cols, rows)
q = synth.quant_tables(85)
rng = np.random.default_rng(0)
cy = rng.integers(-20, 21, size=n * cols * rows * 4 * 64).astype(np.int16)
cu = rng.integers(-20, 21, size=n * cols * rows * 64).astype(np.int16)
cv = cu.copy()
ops.jpeg_recon_batch_host(g, 1, cy[: cols * rows * 256], cu[: cols * rows * 64], cv[: cols * rows * 64], q)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    out = ops.jpeg_recon_batch_host(g, n, cy, cu, cv, q)
    best = min(best, time.perf_counter() - t0)
px = n * g.width * g.height
print(json.dumps({"images": n, "coded": [g.width, g.height], "ms": round(best * 1e3, 2), "Gpx/s": round(px / best / 1e9, 3),
                  "GB/s_over_PCIe": round(px * 7 / best / 1e9, 2)}))
