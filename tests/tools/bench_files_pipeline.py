#!/usr/bin/env python3
"""End to end, files in -> BGRA out in host memory (ffhip_jpeg_decode_files): Huffman decode on host threads
overlapped with H2D + reconstruction + D2H.  Entropy- and PCIe-inclusive: never bench.py's `value`.
Needs PIL to make the 4K test file (restart interval = one MCU row)."""
import io, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from ffpic_amd import capi, ops

capi.require_device(0)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:2160, 0:3840]
img = np.stack([128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), 128 + 90 * np.cos(xx / 11.0 + yy / 53.0), (xx * 255 / 3839 + yy * 255 / 2159) / 2], axis=2)
img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
bio = io.BytesIO(); Image.fromarray(img).save(bio, "JPEG", quality=85, subsampling=2, restart_marker_rows=1)
data = bio.getvalue()
n = 128
files = [data] * n
out = {"file_bytes": len(data), "pictures": n, "coded": [3840, 2160]}
pageable = np.zeros((n, 2160, 3840, 4), np.uint8)     # touched once: no first-touch page faults in the timing
pinned = ops.PinnedArray((n, 2160, 3840, 4))
ops.jpeg_decode_files(files, n_threads=8, chunk=4, out=pageable)
for dst_name, dst, gpu in (("pageable", pageable, "0"), ("pinned", pinned.array, "0"), ("pinned_gpu_entropy", pinned.array, "1"), ("pageable_gpu_entropy", pageable, "1")):
    capi.setenv("FFHIP_JPEG_GPU_ENTROPY", gpu)
    for th in ((1, 8, 16) if gpu == "0" else (4, 16)):
        for chunk in ((4, 8) if gpu == "0" else (16, 32, 64)):
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter(); g, px = ops.jpeg_decode_files(files, n_threads=th, chunk=chunk, out=dst); best = min(best, time.perf_counter() - t0)
            out[f"{dst_name}_threads_{th}_chunk_{chunk}"] = {"ms": round(best * 1e3, 1), "Gpx/s": round(n * g.width * g.height / best / 1e9, 2), "files/s": round(n / best, 1)}
assert np.array_equal(pageable, pinned.array)
capi.setenv("FFHIP_JPEG_GPU_ENTROPY", "0")
# the unpipelined two-step path for comparison
best = 1e9
for _ in range(2):
    t0 = time.perf_counter(); ops.decode_jpeg_files(files, n_threads=16); best = min(best, time.perf_counter() - t0)
out["two_step_threads_16"] = {"ms": round(best * 1e3, 1), "Gpx/s": round(n * 3840 * 2160 / best / 1e9, 2)}
print(json.dumps(out, indent=1))
