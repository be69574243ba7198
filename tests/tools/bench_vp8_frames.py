#!/usr/bin/env python3
"""The VP8 frame chain (ffhip_vp8_decode_frames) by form: the fused frame kernel at several waves-per-frame settings against the
three-stage row form, per batch size and mode source.  SIZES=16,64,256,1024  SOURCES=encoder,random  WAVES=8,16"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from ffpic_amd import capi
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
sizes = tuple(int(x) for x in os.environ.get("SIZES", "16,64,256,1024").split(","))
sources = os.environ.get("SOURCES", "encoder,random").split(",")
waves = os.environ.get("WAVES", "8,16").split(",")
T = bench.Timer(L, st)
X = bench.C4(L, dev, st, T)
out = []
for source in sources:
    for nf in sizes:
        B = X.batch(nf, source)
        reps = 5 if nf <= 64 else 3
        row = {"source": source, "frames": nf}
        capi.setenv("FFHIP_VP8_FRAMES", "rows")
        row["rows_ms"] = round(T.ms(B.s_frames, reps=reps, warm=1), 4)
        for w in waves:
            capi.setenv("FFHIP_VP8_FRAMES", "fused"); capi.setenv("FFHIP_VP8_FRAME_WAVES", w)
            row[f"fused_w{w}_ms"] = round(T.ms(B.s_frames, reps=reps, warm=1), 4)
        capi.setenv("FFHIP_VP8_FRAME_WAVES", None)
        row["residual_ms"] = round(T.ms(B.s_res, reps=reps, warm=1), 4)
        best = min(v for k, v in row.items() if k.endswith("_ms") and k != "residual_ms")
        row["best_chain_Gpx_s"] = round(nf * B.Hp * B.Wp / (best + row["residual_ms"]) / 1e6, 1)
        capi.setenv("FFHIP_VP8_FRAMES", "fused")
        B.s_frames(); capi.check(L.ffhip_stream_sync(st))
        row["parity_fused"] = X.parity(B, source, sorted({0, nf - 1}))
        capi.setenv("FFHIP_VP8_FRAMES", None)
        out.append(row)
        print(json.dumps(row), flush=True)
        del B
        torch.cuda.empty_cache()
