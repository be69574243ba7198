#!/usr/bin/env python3
"""VP8 prediction + loop filter side by side on FRAMES copies of the encoder's frame: a grid over the two wave counts and the
blocking slack (FFHIP_VP8_SLACK)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from ffpic_amd import capi
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
T = bench.Timer(L, st)
X = bench.C4(L, dev, st, T)
out = {}
for nf in [int(x) for x in os.environ.get("FRAMES", "256").split(",")]:
    B = X.batch(nf, os.environ.get("SOURCE", "encoder"))
    B.s_res()
    o = {}
    for slack in (0, 4, 8, 16, 32):
        capi.setenv("FFHIP_VP8_SLACK", slack)
        for pw in (1024, 1536, 2048, 3072):
            capi.setenv("FFHIP_VP8_PRED_WAVES", pw)
            o[f"pred s{slack} w{pw}"] = round(T.ms(B.s_pred, reps=3, warm=1), 3)
        capi.setenv("FFHIP_VP8_PRED_WAVES", None)
        for lw in (512, 1024, 2048):
            capi.setenv("FFHIP_VP8_LF_WAVES", lw)
            o[f"lf s{slack} w{lw}"] = round(T.ms(B.s_lf, reps=3, warm=1), 3)
        for pw, lw in ((1536, 512), (1536, 1024), (2048, 512), (2048, 1024), (2048, 2048), (3072, 1024)):
            capi.setenv("FFHIP_VP8_PRED_WAVES", pw); capi.setenv("FFHIP_VP8_LF_WAVES", lw)
            o[f"fused s{slack} {pw}+{lw}"] = round(T.ms(B.s_pred_lf, reps=3, warm=1), 3)
        capi.setenv("FFHIP_VP8_PRED_WAVES", None); capi.setenv("FFHIP_VP8_LF_WAVES", None)
    capi.setenv("FFHIP_VP8_SLACK", None)
    out[nf] = o
    del B
    torch.cuda.empty_cache()
print(json.dumps(out))
