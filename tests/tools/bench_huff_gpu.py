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
# sustained: T caller threads, each with its own stream, decoding batches of 128 4K files (MCU-row intervals) to device pixels
import threading
bio = io.BytesIO(); Image.fromarray(img).save(bio, "JPEG", quality=85, subsampling=2, restart_marker_rows=1); data = bio.getvalue()
nb = 128
files = [data] * nb
bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
ptrs = (C.c_void_p * nb)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * nb)(*[b.size for b in bufs])
for T in (1, 2, 3):
    streams = [L.ffhip_stream_create() for _ in range(T)]
    douts = [ops.DeviceBuffer(nbytes=nb * g.width * g.height * 4) for _ in range(T)]
    reps = 6
    def worker(t):
        status = (C.c_int * nb)(); gg = capi.JpegGeom()
        for _ in range(reps):
            capi.check(L.ffhip_jpeg_decode_files_device(ptrs, lens, nb, 8, C.byref(gg), douts[t].ptr, g.width * 4, g.width * 4 * g.height, status, streams[t]))
        capi.check(L.ffhip_stream_sync(streams[t]))
    worker(0)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    out[f"sustained_files_to_device_{T}_callers_x_{nb}_files"] = {"files/s": round(T * reps * nb / dt), "Gpx/s": round(T * reps * nb * g.width * g.height / dt / 1e9, 1)}
    for st_ in streams: L.ffhip_stream_destroy(st_)
    del douts
# thumbnails WITHOUT restart markers: 4096 x 256x256, one lane per file on the device vs 16 host threads
small = Image.fromarray(img[:256, :256])
bio = io.BytesIO(); small.save(bio, "JPEG", quality=85, subsampling=2); data = bio.getvalue(); n = 4096
gs, _, _ = ops.jpeg_probe(data)
files = [data] * n
bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
dout = ops.DeviceBuffer(nbytes=n * gs.width * gs.height * 4); status = (C.c_int * n)(); g2 = capi.JpegGeom()
for mode in ("0", "1"):
    capi.setenv("FFHIP_JPEG_GPU_ENTROPY", mode)
    def run3():
        capi.check(L.ffhip_jpeg_decode_files_device(ptrs, lens, n, 16, C.byref(g2), dout.ptr, gs.width * 4, gs.width * 4 * gs.height, status, None))
        capi.check(L.ffhip_stream_sync(None))
    run3()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run3(); best = min(best, time.perf_counter() - t0)
    out[f"thumbnails_256x256_no_dri_n{n}_{'device' if mode == '1' else 'host16'}_entropy"] = {"file_bytes": len(data), "ms": round(best * 1e3, 2), "Gpx/s": round(n * gs.width * gs.height / best / 1e9, 2), "files/s": round(n / best)}
capi.setenv("FFHIP_JPEG_GPU_ENTROPY", None)
print(json.dumps(out, indent=1))
