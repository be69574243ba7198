#!/usr/bin/env python3
"""ffhip_hevc_intra_recon on one 7680x4352 picture, SURVEY 8d's config-5 TU mix and the random quadtree, by scheduling
window (FFHIP_HEVC_INTRA_WINDOW): HIP events around 5 back-to-back calls (planner kernels included)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])   # A/B runs of two builds in one gpurun call
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
W, H = 7680, 4352
out = {}
for tag, mix, seed in [m for m in (("c5mix", "c5", 5), ("quadtree", None, 2)) if m[0] in os.environ.get("MIXES", "c5mix,quadtree").split(",")]:
    tus, res = synth.hevc_intra_tus(W, H, seed=seed, tu_mix=mix)
    dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev); dr = torch.from_numpy(res).to(dev)
    py = torch.zeros((H, W), dtype=torch.int16, device=dev); pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev); pv = torch.zeros_like(pu)
    def run():
        capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), len(tus), dr.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st))
    for wl in sys.argv[1:] or ["6", "5", "4"]:
      for form in os.environ.get("FORMS", "default").split(","):     # FORMS=0,1 with a build of tests/tools/experiments/r3_hevc_intra_*.patch: one wave per group / four waves per group
        capi.setenv("FFHIP_HEVC_INTRA_WINDOW", wl)
        capi.setenv("FFHIP_HEVC_INTRA_FORM4", None if form == "default" else form)
        run(); capi.check(L.ffhip_stream_sync(st))
        L.ffhip_event_record(e0, st)
        for _ in range(5): run()
        L.ffhip_event_record(e1, st)
        capi.check(L.ffhip_stream_sync(st))
        out[f"{tag}_w{1 << int(wl)}" + ("" if form == "default" else f"_form4={form}")] = {"ms": round(L.ffhip_event_elapsed_ms(e0, e1) / 5, 3), "tus": int(len(tus))}
print(json.dumps(out))
