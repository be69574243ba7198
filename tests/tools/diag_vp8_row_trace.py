#!/usr/bin/env python3
"""When the rows of ONE frame start and finish in k_vp8_predict_rows (the real encoder's 1080p frame): per row the 100 MHz clock
at its ticket, at its first macroblock (the first fetch has returned), at its middle macroblock and at its last store.
Prints how long a row takes, how far its end trails the end of the row above, and how long it waited for its first fetch."""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi
dev = torch.device("cuda", 0)
L = capi.require_device(0)
L.ffhip_debug_vp8_trace.argtypes = [C.c_void_p]; L.ffhip_debug_vp8_trace.restype = None
st = torch.cuda.current_stream().cuda_stream
g = np.load(os.path.join(ROOT, "tests", "golden", "webp_file_1080p.npz"))
c, r = 120, 68
n_mb = c * r
modes = np.ascontiguousarray(g["modes"])[None]
res = torch.from_numpy(np.ascontiguousarray(g["residual"])).to(dev); dm = torch.from_numpy(modes).to(dev)
Y = torch.zeros((1, 16 * r, 16 * c), dtype=torch.uint8, device=dev); U = torch.zeros((1, 8 * r, 8 * c), dtype=torch.uint8, device=dev); V = torch.zeros_like(U)
tr = torch.zeros((r, 8), dtype=torch.int64, device=dev)
fn = lambda: capi.check(L.ffhip_vp8_predict_recon(c, r, 1, modes.ctypes.data, dm.data_ptr(), res.data_ptr(), n_mb * 384, None, Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, st))
for _ in range(3): fn()
L.ffhip_debug_vp8_trace(tr.data_ptr())
fn(); capi.check(L.ffhip_stream_sync(st))
L.ffhip_debug_vp8_trace(None)
t = tr.cpu().numpy().astype(np.float64)[:, :4] / 100.0          # us (words 4-7: per-phase sums, see ffhip_vp8_pred.hip)
t -= t[:, 0].min()
dur = t[:, 3] - t[:, 1]; first_half = t[:, 2] - t[:, 1]; second_half = t[:, 3] - t[:, 2]
lag_end = np.diff(t[:, 3]); lag_start = np.diff(t[:, 1]); lag_mid = np.diff(t[:, 2])
bshare = (g["modes"][:, 0].reshape(r, c) == 4).mean(axis=1)
print(json.dumps({"frame_us": round(float(t[:, 3].max()), 1), "row_us_mean": round(float(dur.mean()), 1), "row_us_min_max": [round(float(dur.min()), 1), round(float(dur.max()), 1)],
                  "first_half_us_mean": round(float(first_half.mean()), 1), "second_half_us_mean": round(float(second_half.mean()), 1),
                  "start_after_row_above_us_mean": round(float(lag_start.mean()), 2), "middle_after_row_above_us_mean": round(float(lag_mid.mean()), 2), "end_after_row_above_us_mean": round(float(lag_end.mean()), 2),
                  "ticket_to_first_mb_us_mean": round(float((t[:, 1] - t[:, 0]).mean()), 1), "row0_us": round(float(dur[0]), 1), "row0_b_pred_share": round(float(bshare[0]), 2),
                  "b_pred_share_mean": round(float(bshare.mean()), 2)}))
print("row: start  mid  end  (us) | b_pred share")
for y in (0, 1, 2, 3, 10, 20, 33, 34, 50, 66, 67):
    print(y, [round(float(v), 1) for v in t[y, 1:]], round(float(bshare[y]), 2))
