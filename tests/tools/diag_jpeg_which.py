#!/usr/bin/env python3
"""Which buffer's placement sets the headline kernel's mode?  Re-allocates ONE of (ty, tu, tv, out) at a time (behind a
dummy of varying size), keeps the others, and times 25 launches (mean of the last 10)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
n = 256
cols, rows = 240, 135
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
mcus = cols * rows
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
def mk(name):
    if name == "ty": return torch.randint(-30, 31, (n * mcus * 4, 64), device=dev, dtype=torch.int16)
    if name in ("tu", "tv"): return torch.randint(-30, 31, (n * mcus, 64), device=dev, dtype=torch.int16)
    return torch.empty(n * W * 4 * H, dtype=torch.uint8, device=dev)
B = {k: mk(k) for k in ("ty", "tu", "tv", "out")}
def measure():
    if os.environ.get("VARIANTS"):
        r = []
        for v in os.environ["VARIANTS"].split(","):
            capi.setenv("FFHIP_JPEG_VARIANT", v)
            r.append(measure1())
        return r
    return measure1()
def measure1():
    def step():
        ops.jpeg_recon_batch(geom, n, B["ty"].data_ptr(), B["tu"].data_ptr(), B["tv"].data_ptr(), q.data_ptr(), 0, B["out"].data_ptr(), W * 4, W * 4 * H, None, 0, st)
    ts = []
    for _ in range(25):
        L.ffhip_event_record(e0, st); step(); L.ffhip_event_record(e1, st)
        ts.append(L.ffhip_event_elapsed_ms(e0, e1))
    return round(sum(ts[-10:]) / 10, 3)
print("start", measure(), {k: hex(v.data_ptr()) for k, v in B.items()}, flush=True)
trial = 0
for which in os.environ.get("WHICH", "out,ty,tu,tv,out,ty").split(","):
    for rep in range(5):
        trial += 1
        old = B[which]; B[which] = None; del old
        torch.cuda.empty_cache()
        dummy = torch.empty(1 + (trial % 7) * 37_000_003, dtype=torch.uint8, device=dev)
        B[which] = mk(which)
        del dummy
        print(which, rep, measure(), hex(B[which].data_ptr()), flush=True)
