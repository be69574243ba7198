#!/usr/bin/env python3
"""What ffhip_hevc_intra_recon costs the HOST per 8K picture (list validation, window choice, kernel launches: the call
only enqueues) next to the time until the picture is done: 1.0 of 6.5 ms for the config-5 mix, 2.1 of 12.8 for the quadtree."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
W, H = 7680, 4352
for tag, mix, seed in (("c5mix", "c5", 5), ("quadtree", None, 2)):
    tus, res = synth.hevc_intra_tus(W, H, seed=seed, tu_mix=mix)
    dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev); dr = torch.from_numpy(res).to(dev)
    py = torch.zeros((H, W), dtype=torch.int16, device=dev); pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev); pv = torch.zeros_like(pu)
    def run():
        capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), len(tus), dr.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st))
    run(); capi.check(L.ffhip_stream_sync(st))
    t0 = time.perf_counter(); run(); t1 = time.perf_counter(); capi.check(L.ffhip_stream_sync(st)); t2 = time.perf_counter()
    print(tag, "host enqueue %.2f ms, until done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
