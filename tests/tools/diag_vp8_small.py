#!/usr/bin/env python3
"""ffhip_vp8_predict_recon on a few small pictures against the oracle: C R N [seed].  A smoke run for work on the row kernel (run it under
`timeout`: a kernel that never ends shows here first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ffpic_amd import capi, ops, synth
import oracle_lib as O
c, r, n = (int(x) for x in sys.argv[1:4])
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
modes = np.stack([synth.vp8_modes(c, r, seed=seed + i) for i in range(n)])
resid = np.stack([synth.vp8_residual(c * r, seed=seed + 100 + i) for i in range(n)])
print("calling", c, r, n, flush=True)
y, u, v = ops.vp8_predict_recon(c, r, modes, resid)
print("returned", flush=True)
ok = True
for i in range(n):
    ey, eu, ev = O.oracle_vp8_frame(c, r, modes[i], resid[i])
    for got, exp, name in ((y[i], ey, "Y"), (u[i], eu, "U"), (v[i], ev, "V")):
        if not np.array_equal(got, exp):
            ys_, xs_ = np.nonzero(got != exp)
            print(f"image {i} plane {name}: {len(ys_)} samples differ, first at x={xs_[0]} y={ys_[0]}")
            ok = False
print("equal to the oracle:", ok)
