#!/usr/bin/env python3
"""Does the headline kernel's time depend on WHERE its buffers sit?  One process, the bench's 256 x 4K batch, buffers
re-allocated several times behind dummies of different sizes; every launch timed on its own with HIP events.
Diagnostic for the 0.74-0.80 spread of roofline.frac between runs."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
if os.environ.get("FFHIP_LIB"): capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", os.environ["FFHIP_LIB"])
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
n = int(os.environ.get("FFHIP_BENCH_IMAGES", "256"))
cols, rows = 240, 135
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
mcus = cols * rows
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
res = []
for trial in range(int(os.environ.get("TRIALS", "8"))):
    dummy = torch.empty(1 + trial * 37_000_003, dtype=torch.uint8, device=dev)
    ty = torch.randint(-30, 31, (n * mcus * 4, 64), device=dev, dtype=torch.int16)
    tu = torch.randint(-30, 31, (n * mcus, 64), device=dev, dtype=torch.int16)
    tv = torch.randint(-30, 31, (n * mcus, 64), device=dev, dtype=torch.int16)
    out = torch.empty(n * W * 4 * H, dtype=torch.uint8, device=dev)
    def step():
        ops.jpeg_recon_batch(geom, n, ty.data_ptr(), tu.data_ptr(), tv.data_ptr(), q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H, None, 0, st)
    for _ in range(3): step()
    ts = []
    for _ in range(int(os.environ.get("LAUNCHES", "10"))):
        L.ffhip_event_record(e0, st); step(); L.ffhip_event_record(e1, st)
        ts.append(L.ffhip_event_elapsed_ms(e0, e1))
    def ev_ms(fn, reps=4):
        fn(); L.ffhip_event_record(e0, st)
        for _ in range(reps): fn()
        L.ffhip_event_record(e1, st)
        return L.ffhip_event_elapsed_ms(e0, e1) / reps
    o64 = out.view(torch.int64)
    half = o64.numel() // 2
    w_ms = ev_ms(lambda: out.zero_())                              # write-only stream over the output buffer
    r_ms = ev_ms(lambda: ty.view(torch.int32).bitwise_and_(-1))    # read + write in place over the luma coefficients
    c_ms = ev_ms(lambda: o64[half:half * 2].copy_(o64[:half]))     # copy inside the output buffer
    res.append({"write_TBps": round(out.numel() / w_ms / 1e9, 3), "rw_TBps": round(2 * ty.numel() * 2 / r_ms / 1e9, 3), "copy_TBps": round(2 * half * 8 / c_ms / 1e9, 3), "trial": trial, "min_ms": round(min(ts), 4), "mean_ms": round(sum(ts) / len(ts), 4), "max_ms": round(max(ts), 4),
                "TB/s_mean": round(7 * n * W * H / (sum(ts) / len(ts)) / 1e9, 3),
                "addr": [hex(t.data_ptr()) for t in (ty, tu, tv, out)]})
    if os.environ.get("SERIES"): res[-1]["series"] = [round(t, 3) for t in ts]
    print(json.dumps(res[-1]), flush=True)
    del dummy, ty, tu, tv, out
    torch.cuda.empty_cache()
