#!/usr/bin/env python3
"""Sustained files -> device pixels: T threads (default 2), each with a stream and buffers of its own, decode batches of N (256) copies of bench.py's 4K
file (RESTART_ROWS: with restart markers) back to back for ROUNDS (12) calls each; one caller's host work -- header parsing, staging -- runs under the
other's uploads and kernels.  Prints calls, wall time and the aggregate rate."""
import ctypes as C, io, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from PIL import Image
from ffpic_amd import capi

T = int(os.environ.get("T", 2)); n = int(os.environ.get("N", 256)); rounds = int(os.environ.get("ROUNDS", 12)); host_threads = int(os.environ.get("HOST_THREADS", 8))
W, H = 3840, 2160
rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:H, 0:W]
img = np.stack([128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), 128 + 90 * np.cos(xx / 11.0 + yy / 53.0), (xx * 255 / (W - 1) + yy * 255 / (H - 1)) / 2], axis=2)
img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
kw = {"restart_marker_rows": int(os.environ["RESTART_ROWS"])} if os.environ.get("RESTART_ROWS") else {}
bio = io.BytesIO(); Image.fromarray(img).save(bio, "JPEG", quality=85, subsampling=2, **kw); data = bio.getvalue()
L = capi.require_device(); dev = torch.device("cuda:0")
buf = np.frombuffer(data, dtype=np.uint8)
go = threading.Barrier(T + 1)
errors, done_at = [], [0.0] * T

def worker(k):
    try:
        ptrs = (C.c_void_p * n)(*([buf.ctypes.data] * n)); lens = (C.c_size_t * n)(*([buf.size] * n)); status = (C.c_int * n)()
        g = capi.JpegGeom()
        st = torch.cuda.Stream(device=dev)
        out = torch.empty((n, H, W * 4), dtype=torch.uint8, device=dev)
        def call():
            capi.check(L.ffhip_jpeg_decode_files_device(ptrs, lens, n, host_threads, C.byref(g), out.data_ptr(), W * 4, W * 4 * H, status, st.cuda_stream))
            capi.check(L.ffhip_stream_sync(st.cuda_stream))
        call()                                   # buffers, streams
        go.wait()
        for _ in range(rounds):
            call()
        done_at[k] = time.perf_counter()
    except Exception as e:      # noqa: BLE001
        errors.append(repr(e))
        try:
            go.abort()
        except Exception:       # noqa: BLE001
            pass

threads = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
for t in threads:
    t.start()
go.wait()
t0 = time.perf_counter()
for t in threads:
    t.join()
if errors:
    sys.exit("; ".join(errors))
dt = max(done_at) - t0
print(json.dumps({"threads": T, "host_threads_each": host_threads, "files_per_call": n, "calls": T * rounds, "seconds": round(dt, 4), "ms_per_call_aggregate": round(dt * 1e3 / (T * rounds), 2),
                  "files_per_s": round(T * rounds * n / dt), "Mpixels_per_s": round(T * rounds * n * W * H / dt / 1e6, 1)}))
