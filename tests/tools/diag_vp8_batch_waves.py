#!/usr/bin/env python3
"""VP8 prediction / loop filter on FRAMES copies of the encoder's frame: time against the number of waves in the launch
(FFHIP_VP8_PRED_WAVES / FFHIP_VP8_LF_WAVES), alone and side by side, plus the per-row trace of the prediction kernel
(how long a row takes under load, how long it waits between its ticket and its first macroblock)."""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from ffpic_amd import capi
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])
dev = torch.device("cuda", 0)
L = capi.require_device(0)
L.ffhip_debug_vp8_trace.argtypes = [C.c_void_p]; L.ffhip_debug_vp8_trace.restype = None
st = torch.cuda.current_stream().cuda_stream
T = bench.Timer(L, st)
X = bench.C4(L, dev, st, T)
nf = int(os.environ.get("FRAMES", "256"))
B = X.batch(nf, os.environ.get("SOURCE", "encoder"))
B.s_res()
out = {"frames": nf, "pred": {}, "lf": {}, "fused": {}}
for w in [int(x) for x in os.environ.get("PRED_WAVES", "512,1024,2048,3072,4096,8192").split(",")]:
    capi.setenv("FFHIP_VP8_PRED_WAVES", w)
    out["pred"][w] = round(T.ms(B.s_pred, reps=3, warm=1), 3)
capi.setenv("FFHIP_VP8_PRED_WAVES", None)
for w in (1024, 2048, 3584, 7168):
    capi.setenv("FFHIP_VP8_LF_WAVES", w)
    out["lf"][w] = round(T.ms(B.s_lf, reps=3, warm=1), 3)
capi.setenv("FFHIP_VP8_LF_WAVES", None)
for pw, lw in ((2048, 3584), (2048, 2048), (3072, 1024), (3072, 2048), (3584, 512), (4096, 1024)):
    capi.setenv("FFHIP_VP8_PRED_WAVES", pw); capi.setenv("FFHIP_VP8_LF_WAVES", lw)
    out["fused"][f"{pw}+{lw}"] = round(T.ms(B.s_pred_lf, reps=3, warm=1), 3)
capi.setenv("FFHIP_VP8_PRED_WAVES", None); capi.setenv("FFHIP_VP8_LF_WAVES", None)
# row trace of the prediction kernel alone, default wave count
r, c = X.r, X.c
tr = torch.zeros((nf * r, 8), dtype=torch.int64, device=dev)
L.ffhip_debug_vp8_trace(tr.data_ptr())
B.s_pred(); capi.check(L.ffhip_stream_sync(st))
L.ffhip_debug_vp8_trace(None)
raw = tr.cpu().numpy().astype(np.float64).reshape(nf, r, 8) / 100.0
t = raw[..., :4].copy()
ph = raw[..., 4:]
t -= t[..., 0].min()
dur = t[..., 3] - t[..., 1]
wait = t[..., 1] - t[..., 0]
out["trace"] = {"kernel_us": round(float(t[..., 3].max()), 1), "row_us_mean": round(float(dur.mean()), 1), "row_us_p10_p50_p90": [round(float(np.percentile(dur, p)), 1) for p in (10, 50, 90)],
                "first_half_us_mean": round(float((t[..., 2] - t[..., 1]).mean()), 1), "second_half_us_mean": round(float((t[..., 3] - t[..., 2]).mean()), 1),
                "ticket_to_first_mb_us_mean": round(float(wait.mean()), 1), "ticket_to_first_mb_p50_p90": [round(float(np.percentile(wait, p)), 1) for p in (50, 90)],
                "phase_us_per_row_mean": {"fetch_wait+consume": round(float(ph[..., 0].mean()), 1), "poll+issue_next_fetch": round(float(ph[..., 1].mean()), 1), "luma": round(float(ph[..., 2].mean()), 1), "chroma": round(float(ph[..., 3].mean()), 1)},
                "phase_us_row0_mean": [round(float(ph[:, 0, k].mean()), 1) for k in range(4)],
                "row_us_by_row_index": [round(float(dur[:, y].mean()), 1) for y in (0, 1, 7, 8, 9, 15, 16, 33, 66, 67)]}
print(json.dumps(out))
