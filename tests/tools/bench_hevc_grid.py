#!/usr/bin/env python3
"""bench.py's `extra.c5.grid` on its own: 8K pictures as grids of 135 independent 512x512 HEVC tiles, 1 / 4 / 8 pictures per call.
PICTURES=1,4,8  FFHIP_HEVC_INTRA_WAVES=<n>  FFHIP_LIB=<other build>"""
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
pics = tuple(int(x) for x in os.environ.get("PICTURES", "1,4,8").split(","))
print(json.dumps(bench.c5_grid_sweep(L, dev, st, bench.Timer(L, st), cpu=not os.environ.get("NO_CPU"), pictures=pics)))
