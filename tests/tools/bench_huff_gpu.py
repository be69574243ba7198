#!/usr/bin/env python3
"""The entropy front end on the device (ffhip_jpeg_entropy_batch_gpu, one lane per restart interval) on 4K JPEG
files with one restart interval per MCU row, next to the host threads.  Wall time of the call (host header parse +
marker scan + upload + kernel + verdict read-back); kernel time alone comes from rocprofv3 --kernel-trace."""
import io, os, sys, time, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from ffpic_amd import capi, ops

L = capi.require_device(0)
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:2160, 0:3840]
img = np.stack([128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), 128 + 90 * np.cos(xx / 11.0 + yy / 53.0), (xx * 255 / 3839 + yy * 255 / 2159) / 2], axis=2)
img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
out = {}
for rows in (1, 0):     # 0 -> restart every 16 MCUs instead
    bio = io.BytesIO()
    kw = dict(restart_marker_rows=1) if rows else dict(restart_marker_blocks=16)
    Image.fromarray(img).save(bio, "JPEG", quality=85, subsampling=2, **kw)
    data = bio.getvalue()
    g, _, _ = ops.jpeg_probe(data)
    for n in (16, 64, 256):
        files = [data] * n
        bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
        dy = ops.DeviceBuffer(nbytes=n * g.y_blocks * 128); du = ops.DeviceBuffer(nbytes=n * g.c_blocks * 128); dv = ops.DeviceBuffer(nbytes=n * g.c_blocks * 128)
        dq = ops.DeviceBuffer(nbytes=n * 512); status = (C.c_int * n)()
        def run():
            capi.check(L.ffhip_jpeg_entropy_batch_gpu(ptrs, lens, n, 16, C.byref(g), dy.ptr, du.ptr, dv.ptr, dq.ptr, status, None))
        run()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
        out[f"{'row' if rows else '16mcu'}_intervals_n{n}"] = {"file_bytes": len(data), "ms": round(best * 1e3, 2), "Gpx/s": round(n * g.width * g.height / best / 1e9, 2), "files/s": round(n / best)}
# files -> BGRA in DEVICE memory (entropy on the device + reconstruction), 256 x 4K
for rows in (1, 0):
    bio = io.BytesIO()
    kw = dict(restart_marker_rows=1) if rows else dict(restart_marker_blocks=16)
    Image.fromarray(img).save(bio, "JPEG", quality=85, subsampling=2, **kw)
    data = bio.getvalue(); n = 256
    files = [data] * n
    bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
    dout = ops.DeviceBuffer(nbytes=n * g.width * g.height * 4); status = (C.c_int * n)(); g2 = capi.JpegGeom()
    def run2():
        capi.check(L.ffhip_jpeg_decode_files_device(ptrs, lens, n, 16, C.byref(g2), dout.ptr, g.width * 4, g.width * 4 * g.height, status, None))
        capi.check(L.ffhip_stream_sync(None))
    run2()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run2(); best = min(best, time.perf_counter() - t0)
    out[f"files_to_device_pixels_{'row' if rows else '16mcu'}_intervals_n{n}"] = {"ms": round(best * 1e3, 2), "Gpx/s": round(n * g.width * g.height / best / 1e9, 2), "files/s": round(n / best)}
    del dout
print(json.dumps(out, indent=1))
