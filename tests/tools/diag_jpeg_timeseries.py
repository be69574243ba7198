#!/usr/bin/env python3
"""Headline kernel, ONE allocation, several seconds of back-to-back launches, each timed with its own pair of HIP
events: does the kernel's time move with TIME rather than with placement?  Prints the mean of every 25 launches."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
n = int(os.environ.get("FFHIP_BENCH_IMAGES", "256"))
N = int(os.environ.get("LAUNCHES", "2000"))
GAP = float(os.environ.get("GAP_MS", "0"))      # idle time between launches
cols, rows = 240, 135
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
mcus = cols * rows
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
ty = torch.randint(-30, 31, (n * mcus * 4, 64), device=dev, dtype=torch.int16)
tu = torch.randint(-30, 31, (n * mcus, 64), device=dev, dtype=torch.int16)
tv = torch.randint(-30, 31, (n * mcus, 64), device=dev, dtype=torch.int16)
out = torch.empty(n * W * 4 * H, dtype=torch.uint8, device=dev)
evs = [L.ffhip_event_create() for _ in range(N + 1)]
def step():
    ops.jpeg_recon_batch(geom, n, ty.data_ptr(), tu.data_ptr(), tv.data_ptr(), q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H, None, 0, st)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.time()
if GAP == 0:
    L.ffhip_event_record(evs[0], st)
    for i in range(N):
        step(); L.ffhip_event_record(evs[i + 1], st)
    torch.cuda.synchronize()
    ts = [L.ffhip_event_elapsed_ms(evs[i], evs[i + 1]) for i in range(N)]
else:
    ts = []
    for i in range(N):
        L.ffhip_event_record(evs[0], st); step(); L.ffhip_event_record(evs[1], st); torch.cuda.synchronize()
        ts.append(L.ffhip_event_elapsed_ms(evs[0], evs[1])); time.sleep(GAP / 1e3)
print("wall_s", round(time.time() - t0, 2), "launches", N, "gap_ms", GAP)
B = 25
print(" ".join(f"{sum(ts[i:i + B]) / B:.3f}" for i in range(0, N, B)))
print("min", round(min(ts), 4), "max", round(max(ts), 4), "mean", round(sum(ts) / N, 4))
