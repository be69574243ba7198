#!/usr/bin/env python3
"""A few calls of the VP8 prediction + loop filter on FRAMES copies of the encoder's 1080p frame (default 256), for rocprofv3:
  rocprofv3 --kernel-trace --stats -- python3 tests/tools/prof_vp8_batch.py      rocprofv3 --pmc ... -- python3 tests/tools/prof_vp8_batch.py
MODE=fused (default: the two row kernels side by side) | pred | lf | seq | frames (ffhip_vp8_decode_frames: the one-kernel form of round 4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from ffpic_amd import capi
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
X = bench.C4(L, dev, st, bench.Timer(L, st))
nf = int(os.environ.get("FRAMES", "256"))
B = X.batch(nf, os.environ.get("SOURCE", "encoder"))
B.s_res()
mode = os.environ.get("MODE", "fused")
for _ in range(int(os.environ.get("CALLS", "3"))):
    if mode == "fused": B.s_pred_lf()
    elif mode == "pred": B.s_pred()
    elif mode == "lf": B.s_lf()
    elif mode == "frames": B.s_frames()
    else: B.s_pred(); B.s_lf()
capi.check(L.ffhip_stream_sync(st))
print("done", nf, mode)
