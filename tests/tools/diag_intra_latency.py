#!/usr/bin/env python3
"""Per-TU latency of the grouped HEVC intra kernel: one 64x64 luma block cut into uniform n x n TUs
in z-order = ONE group walked by one wave, so launch time / TU count is the in-group hop cost.
Diagnostic for DESIGN.md section 4; prints one line per (n, mode, flags)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])   # A/B runs of two builds in one gpurun call

dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()


def zorder(x0, y0, size, n, out):
    if size == n:
        out.append((x0, y0))
        return
    h = size // 2
    for dx, dy in ((0, 0), (h, 0), (0, h), (h, h)):
        zorder(x0 + dx, y0 + dy, h, n, out)


def make(n, mode, flags, W=64, H=64):
    pos = []
    for cy in range(0, H, 64):
        for cx in range(0, W, 64):
            zorder(cx, cy, 64, n, pos)
    done = np.zeros((H, W), bool)
    tus, off = [], 0
    for (x0, y0) in pos:
        at = al = 0
        for k in range(2 * n):
            if y0 > 0 and x0 + k < W and done[y0 - 1, x0 + k]: at |= 1 << k
            if x0 > 0 and y0 + k < H and done[y0 + k, x0 - 1]: al |= 1 << k
        fl = flags | (1 if x0 > 0 and y0 > 0 else 0)
        tus.append((x0, y0, int(np.log2(n)), 0, mode, fl, off, 0, at, al))
        off += n * n
        done[y0:y0 + n, x0:x0 + n] = True
    return np.array(tus, dtype=synth.HEVC_TU_DTYPE), np.random.default_rng(0).integers(-20, 20, off).astype(np.int16)


for n in (4, 8, 16, 32):
    for mode, flags, name in ((1, 2, "DC+res"), (0, 2 | 4, "planar+filter"), (34, 2 | 4, "ang34+filter"), (10, 2 | 64, "hor+rdpcm"), (20, 2, "ang20"),
                              (1, 0, "DC nores")):
        tus, res = make(n, mode, flags)
        dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev); dr = torch.from_numpy(res).to(dev)
        py = torch.zeros((64, 64), dtype=torch.int16, device=dev)
        def run():
            capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), len(tus), dr.data_ptr(), py.data_ptr(), None, None, 64, 64, 64, 0, 0, 0, 8, 8, st))
        for _ in range(3): run()
        capi.check(L.ffhip_stream_sync(st))
        L.ffhip_event_record(e0, st)
        for _ in range(20): run()                     # the call only enqueues: planner kernels + the grouped kernel, back to back
        L.ffhip_event_record(e1, st)
        capi.check(L.ffhip_stream_sync(st))
        us = L.ffhip_event_elapsed_ms(e0, e1) / 20 * 1e3
        print(f"n={n:2d} {name:14s} tus={len(tus):4d} {us:8.1f} us per call (planner + kernel) -> {us / len(tus):6.2f} us/TU")
