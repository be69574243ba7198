#!/usr/bin/env python3
"""Per-macroblock latency of the row-form VP8 predictor: ONE row of 512 macroblocks = one wave walking
it, so launch time / 512 is the in-row cost per macroblock, by mode.  Kernel times come from rocprofv3
(--kernel-trace); this prints wall times as an upper bound.  Diagnostic for DESIGN.md section 4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth

dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
c, r = 512, 1
for name, ym, bm in (("DC16", 0, 0), ("TM16", 1, 0), ("V16", 2, 0), ("B_PRED dc", 4, 0), ("B_PRED tm", 4, 1), ("B_PRED taps", 4, 5)):
    modes = np.zeros((1, c * r, 20), np.uint8)
    modes[..., 0] = ym
    modes[..., 2:18] = bm
    resid = torch.from_numpy(synth.vp8_residual(c * r, seed=1)[None]).to(dev)
    dm = torch.from_numpy(modes).to(dev)
    Y = torch.zeros((1, 16 * r, 16 * c), dtype=torch.uint8, device=dev); U = torch.zeros((1, 8 * r, 8 * c), dtype=torch.uint8, device=dev); V = torch.zeros_like(U)
    def run():
        capi.check(L.ffhip_vp8_predict_recon(c, r, 1, modes.ctypes.data, dm.data_ptr(), resid.data_ptr(), c * r * 384, None, Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * c * r, 64 * c * r, st))
    run(); capi.check(L.ffhip_stream_sync(st))
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); run(); capi.check(L.ffhip_stream_sync(st)); best = min(best, (time.perf_counter() - t0) * 1e6)
    print(f"{name:12s} wall {best:8.1f} us -> {best / (c * r):5.2f} us/MB")
