#!/usr/bin/env python3
"""Headline kernel time against the RELATIVE placement of its four buffers inside ONE device allocation (so that their
relative physical offsets are whatever the paddings say, as far as the allocation is physically contiguous)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
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
PITCH_PADS = [int(v) for v in os.environ.get("PITCH_PADS", os.environ.get("PITCH_PAD", "0")).split(",")]   # several: swept inside ONE process (one placement)
STRIDE_PAD = int(os.environ.get("STRIDE_PAD", "0"))
PITCH = W * 4 + max(PITCH_PADS)
ISTRIDE = PITCH * H + STRIDE_PAD
by, bc, bo = n * mcus * 512, n * mcus * 128, n * ISTRIDE
SLACK = int(os.environ.get("SLACK_MB", "64")) << 20
arena = torch.empty(by + 2 * bc + bo + 4 * SLACK, dtype=torch.uint8, device=dev)
arena.view(torch.int16).random_(-30, 31)
base = arena.data_ptr()
def al(x, a=2 << 20): return (x + a - 1) // a * a
pads = [(0, 0, 0), (4096, 8192, 12288), (1 << 16, 2 << 16, 3 << 16), (1 << 20, 0, 1 << 20), (0, 1 << 20, 0), (256, 512, 768),
        (2048, 4096 + 2048, 1024), (1 << 18, 1 << 19, (1 << 18) + (1 << 19)), (1 << 12, 0, 0), (0, 0, 1 << 12), (1 << 14, 1 << 15, (1 << 14) + (1 << 15)), (0, 0, 0)]
if os.environ.get("PADS"): pads = [tuple(int(v) for v in p.split(",")) for p in os.environ["PADS"].split(";")]
print("arena", hex(base), flush=True)
for pp in [(pd, pt) for pt in PITCH_PADS for pd in pads]:
    pp, pt = pp
    PITCH = W * 4 + pt; ISTRIDE = PITCH * H + STRIDE_PAD
    p0, p1, p2, p3 = pp if len(pp) == 4 else (0,) + tuple(pp)
    oy = p0; ou = al(by + SLACK) + p1; ov = al(ou + bc + SLACK) - p1 + p2; oo = al(al(by + SLACK) + 2 * (bc + SLACK) + (4 << 20)) + p3
    def step():
        ops.jpeg_recon_batch(geom, n, base + oy, base + ou, base + ov, q.data_ptr(), 0, base + oo, PITCH, ISTRIDE, None, 0, st)
    for _ in range(3): step()
    ts = []
    for _ in range(8):
        L.ffhip_event_record(e0, st); step(); L.ffhip_event_record(e1, st)
        ts.append(L.ffhip_event_elapsed_ms(e0, e1))
    print(json.dumps({"pitch": PITCH, "pads": [p0, p1, p2, p3], "min_ms": round(min(ts), 4), "mean_ms": round(sum(ts) / len(ts), 4), "TB/s_mean": round(7 * n * W * H / (sum(ts) / len(ts)) / 1e9, 3)}), flush=True)
