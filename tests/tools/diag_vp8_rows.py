#!/usr/bin/env python3
"""What a macroblock row costs alone and what a row adds to a frame: ffhip_vp8_predict_recon on the first r rows of the real
encoder's 1080p frame (r = 1, 2, 4, 8, 16, 34, 68), one frame per call.  r = 1 has no dependency: time / 120 = the mean
macroblock time; (t(r) - t(1)) / (r - 1) = what every further row costs the chain."""
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
g = np.load(os.path.join(ROOT, "tests", "golden", "webp_file_1080p.npz"))
c = 120
out = {}
for r in (1, 2, 4, 8, 16, 34, 68):
    n_mb = c * r
    modes = np.ascontiguousarray(g["modes"][:n_mb])[None]
    res = torch.from_numpy(np.ascontiguousarray(g["residual"][:n_mb])).to(dev)
    dm = torch.from_numpy(modes).to(dev)
    Y = torch.zeros((1, 16 * r, 16 * c), dtype=torch.uint8, device=dev); U = torch.zeros((1, 8 * r, 8 * c), dtype=torch.uint8, device=dev); V = torch.zeros_like(U)
    fn = lambda: capi.check(L.ffhip_vp8_predict_recon(c, r, 1, modes.ctypes.data, dm.data_ptr(), res.data_ptr(), n_mb * 384, None, Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, st))
    for _ in range(3): fn()
    L.ffhip_event_record(e0, st)
    for _ in range(10): fn()
    L.ffhip_event_record(e1, st)
    out[r] = round(L.ffhip_event_elapsed_ms(e0, e1) / 10 * 1e3, 1)
print(json.dumps({"us_by_rows": out, "b_pred_share_row0": round(float((g["modes"][:c, 0] == 4).mean()), 2)}))
