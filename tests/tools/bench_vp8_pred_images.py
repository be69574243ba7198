#!/usr/bin/env python3
"""ffhip_vp8_predict_recon wall time by number of 1080p frames in the call (the frames are independent: if the stage were
bound by one frame's dependency chain alone, 1 and 16 frames would take the same time), and the same with the wrapped H_PRED
of the reference (predict.c:346-353, mode 3 at x = 0 waits for the whole row above) taken out of the synthetic mode mix."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
c, r = 120, 68
out = {}
for tag in ("random modes", "no H_PRED in column 0"):
    for nf in (1, 2, 4, 16):
        modes = np.stack([synth.vp8_modes(c, r, seed=i) for i in range(nf)])
        if tag != "random modes":
            m = modes.reshape(nf, r, c, 20)
            m[:, :, 0, 0] = np.where(m[:, :, 0, 0] == 3, 1, m[:, :, 0, 0])
        resid = torch.from_numpy(np.stack([synth.vp8_residual(c * r, seed=i) for i in range(nf)])).to(dev)
        dm = torch.from_numpy(modes).to(dev)
        Y = torch.zeros((nf, 16 * r, 16 * c), dtype=torch.uint8, device=dev); U = torch.zeros((nf, 8 * r, 8 * c), dtype=torch.uint8, device=dev); V = torch.zeros_like(U)
        def pred():
            capi.check(L.ffhip_vp8_predict_recon(c, r, nf, modes.ctypes.data, dm.data_ptr(), resid.data_ptr(), c * r * 384, None, Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * c * r, 64 * c * r, st))
        pred(); capi.check(L.ffhip_stream_sync(st))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); pred(); capi.check(L.ffhip_stream_sync(st)); best = min(best, (time.perf_counter() - t0) * 1e3)
        out[f"{tag}, {nf} frames"] = round(best, 3)
print(json.dumps(out, indent=1))
