#!/usr/bin/env python3
"""Six back-to-back ffhip_hevc_intra_recon calls on ONE 8K picture (argv[1] = c5 | quadtree), nothing else: the thing to put
under `rocprofv3 --kernel-trace --stats` for the per-picture cost of each kernel of the call (planner, pre-pass, grouped kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
W, H = 7680, 4352
mix = sys.argv[1] if len(sys.argv) > 1 else "c5"
tus, res = synth.hevc_intra_tus(W, H, seed=5 if mix == "c5" else 2, tu_mix="c5" if mix == "c5" else None)
dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev); dr = torch.from_numpy(res).to(dev)
py = torch.zeros((H, W), dtype=torch.int16, device=dev); pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev); pv = torch.zeros_like(pu)
for _ in range(6):
    capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), len(tus), dr.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st))
capi.check(L.ffhip_stream_sync(st))
