#!/usr/bin/env python3
"""ffhip_vp8_predict_loopfilter (the two row kernels side by side) against the two calls one after the other, 16 x 1080p:
the real encoder's frame and uniformly random modes.  HIP events over 10 calls each."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
c, r, nf = 120, 68, 16
n_mb = c * r
g = np.load(os.path.join(ROOT, "tests", "golden", "webp_file_1080p.npz"))
def ms(fn, reps=10):
    for _ in range(2): fn()
    L.ffhip_event_record(e0, st)
    for _ in range(reps): fn()
    L.ffhip_event_record(e1, st)
    return round(L.ffhip_event_elapsed_ms(e0, e1) / reps, 4)
out = {}
Y = torch.zeros((nf, 16 * r, 16 * c), dtype=torch.uint8, device=dev); U = torch.zeros((nf, 8 * r, 8 * c), dtype=torch.uint8, device=dev); V = torch.zeros_like(U)
for tag in ("encoder", "random"):
    if tag == "encoder":
        modes = np.ascontiguousarray(np.broadcast_to(g["modes"], (nf,) + g["modes"].shape))
        res = torch.from_numpy(np.ascontiguousarray(g["residual"])).to(dev).repeat(nf, 1)
        lfv = g["lf"]; filt = lfv[3:27].astype(np.uint8).reshape(4, 2, 3); ft = 1 if lfv[1] else 2
    else:
        modes = np.stack([synth.vp8_modes(c, r, seed=100 + i) for i in range(nf)])
        res = torch.from_numpy(np.stack([synth.vp8_residual(n_mb, seed=i) for i in range(nf)]).reshape(nf * n_mb, 384)).to(dev)
        filt = synth.vp8_filters(seed=2); ft = 2
    dm = torch.from_numpy(modes).to(dev); df = torch.from_numpy(np.ascontiguousarray(filt)).to(dev)
    def seq():
        capi.check(L.ffhip_vp8_predict_recon(c, r, nf, modes.ctypes.data, dm.data_ptr(), res.data_ptr(), n_mb * 384, None, Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, st))
        capi.check(L.ffhip_vp8_loopfilter(c, r, nf, ft, dm.data_ptr(), df.data_ptr(), Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, st))
    def fused():
        capi.check(L.ffhip_vp8_predict_loopfilter(c, r, nf, modes.ctypes.data, dm.data_ptr(), res.data_ptr(), n_mb * 384, None, ft, df.data_ptr(), Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, st))
    out[tag] = {"one_after_the_other_ms": ms(seq), "side_by_side_ms": ms(fused)}
print(json.dumps(out))
