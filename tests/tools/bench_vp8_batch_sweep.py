#!/usr/bin/env python3
"""bench.py's `extra.c4.batch_sweep` on its own: the VP8 chain (residual -> predict || loop filter -> BGRA) against the number
of 1080p frames in one call.  SIZES=1,16,64,256,1024  FFHIP_LIB=<other build>  FFHIP_VP8_FUSE=0  FFHIP_VP8_PRED_WAVES / _LF_WAVES=<n>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from ffpic_amd import capi
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
sizes = tuple(int(x) for x in os.environ.get("SIZES", "1,16,64,256,1024").split(","))
X = bench.C4(L, dev, st, bench.Timer(L, st))
print(json.dumps({"switches": {k: v for k, v in os.environ.items() if k.startswith("FFHIP_")}, **X.sweep(sizes)}))
