#!/usr/bin/env python3
"""ffhip_vp8_predict_loopfilter with 64 and 130 1080p frames in one call (both kernels at their cap of 2 048 waves, far more rows
than waves) against the two calls one after the other, every byte."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ffpic_amd import ops, synth
c, r = 120, 68
for n in (64, 130):
    modes = np.stack([synth.vp8_modes(c, r, seed=900 + (i % 5)) for i in range(n)])
    resid = np.stack([synth.vp8_residual(c * r, seed=910 + (i % 5)) for i in range(n)])
    flt = synth.vp8_filters(seed=3)
    fused = ops.vp8_predict_loopfilter(c, r, modes, resid, 2, flt)
    y0, u0, v0 = ops.vp8_predict_recon(c, r, modes[:5], resid[:5])
    seq = ops.vp8_loopfilter(c, r, 2, modes[:5], flt, y0, u0, v0)
    ok = all(np.array_equal(fused[k][i], seq[k][i % 5]) for k in range(3) for i in range(n))
    print(n, "frames side by side == one after the other:", ok)
