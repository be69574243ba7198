import ctypes as C, io, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from PIL import Image
from ffpic_amd import capi
L = capi.require_device()
rng = np.random.default_rng(1)
base = []
for i in range(64):
    yy, xx = np.mgrid[0:256, 0:256]
    img = np.stack([128 + 100 * np.sin(xx / (9.0 + i)), 128 + 90 * np.cos(yy / (7.0 + i % 5)), (xx * 3 + yy * 5 + i * 7) % 256], axis=2)
    img = np.clip(img + rng.normal(0, 20, img.shape), 0, 255).astype(np.uint8)
    bio = io.BytesIO(); Image.fromarray(img).save(bio, "JPEG", quality=80, subsampling=2); base.append(bio.getvalue())
n = 4096
bufs = [np.frombuffer(base[i % 64], dtype=np.uint8) for i in range(n)]
ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs]); status = (C.c_int * n)()
g = capi.JpegGeom()
W = H = 256
L.ffhip_host_malloc.restype = C.c_void_p
out = L.ffhip_host_malloc(C.c_size_t(n * W * H * 4))
for chunk in (0, 32, 256, 1024, 4096):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        capi.check(L.ffhip_jpeg_decode_files(ptrs, lens, n, 16, chunk, C.byref(g), C.c_void_p(out), C.c_int64(W * 4), C.c_int64(W * 4 * H), status))
        best = min(best, time.perf_counter() - t0)
    print("chunk", chunk, "best ms", round(best * 1e3, 2), "files/s", round(n / best), flush=True)
