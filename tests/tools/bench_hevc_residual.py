#!/usr/bin/env python3
"""k_hevc_residual* alone: four 8K luma planes of TUs per launch (531 MB of traffic, past the Infinity
Cache), HIP events on the launch stream.  FFHIP_HEVC_RES32=dot puts the 32x32 TUs back on the
butterfly kernel for an A/B.  Also a run with a scaling list and mixed flags (the slow paths)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ffpic_amd import capi

dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()

def timeit(fn, reps=20):
    for _ in range(3): fn()
    L.ffhip_event_record(e0, st)
    for _ in range(reps): fn()
    L.ffhip_event_record(e1, st)
    return L.ffhip_event_elapsed_ms(e0, e1) / reps

out = {"FFHIP_HEVC_RES32": os.environ.get("FFHIP_HEVC_RES32", "mfma")}
for n, cnt in ((32, 4 * 240 * 135), (16, 4 * 480 * 270), (8, 4 * 960 * 540), (4, 4 * 1920 * 1080)):
    lvl = torch.randint(-20, 21, (cnt, n * n), device=dev).to(torch.int16)
    info = torch.zeros((cnt, 4), dtype=torch.uint8, device=dev); info[:, 0] = 27
    res = torch.empty_like(lvl)
    sc = torch.randint(1, 256, (6, n * n), device=dev).to(torch.uint8)
    for tag, scp, bd, epp in (("flat_8bit", None, 8, 0), ("list_10bit", sc.data_ptr(), 10, 0), ("flat_12bit_epp", None, 12, 1)):
        ms = timeit(lambda: capi.check(L.ffhip_hevc_residual_batch(n, cnt, lvl.data_ptr(), info.data_ptr(), scp, bd, epp, res.data_ptr(), st)))
        out[f"{n}x{n}_{tag}"] = {"ms": round(ms, 4), "GB/s": round(4 * cnt * n * n / ms / 1e6, 1)}
print(json.dumps(out))
