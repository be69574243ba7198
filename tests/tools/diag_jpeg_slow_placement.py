#!/usr/bin/env python3
"""Find an OUTPUT-buffer placement on which the headline kernel runs in its slow mode, hold it, and time the kernel's
A/B switches on exactly that placement (and on a fast one for comparison): store policy, quads per wave, the XCD mapping of workgroups,
row pitch.  One process, one gpurun call.  SWITCHES="NAME=VALUE,...;NAME=VALUE" adds switch sets."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
n, cols, rows = 256, 240, 135
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
mcus = cols * rows
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
ty = torch.randint(-30, 31, (n * mcus * 256,), device=dev, dtype=torch.int16)
tu = torch.randint(-30, 31, (n * mcus * 64,), device=dev, dtype=torch.int16)
tv = torch.randint(-30, 31, (n * mcus * 64,), device=dev, dtype=torch.int16)
PAD = 8192
import ctypes as C
PATTERN = False   # time ffhip_jpeg_pattern_calibrate (the kernel's loads and stores without the arithmetic) instead of the kernel
def timed(out, pitch, reps=6):
    stride = pitch * H
    def step():
        if PATTERN: capi.check(L.ffhip_jpeg_pattern_calibrate(C.byref(geom), n, ty.data_ptr(), tu.data_ptr(), tv.data_ptr(), q.data_ptr(), 0, out.data_ptr(), pitch, stride, st))
        else: ops.jpeg_recon_batch(geom, n, ty.data_ptr(), tu.data_ptr(), tv.data_ptr(), q.data_ptr(), 0, out.data_ptr(), pitch, stride, None, 0, st)
    for _ in range(2): step()
    L.ffhip_event_record(e0, st)
    for _ in range(reps): step()
    L.ffhip_event_record(e1, st); capi.check(L.ffhip_stream_sync(st))
    ms = L.ffhip_event_elapsed_ms(e0, e1) / reps
    return round(7 * n * W * H / ms / 1e9, 3)
held, slow, fast = [], None, None
seen = []
for attempt in range(14):               # 14 x 12 GB: well inside the card's 288 GB
    out = torch.empty(n * (W * 4 + PAD) * H, dtype=torch.uint8, device=dev)
    t = timed(out, W * 4)
    seen.append((t, out))
    print(json.dumps({"attempt": attempt, "ptr": hex(out.data_ptr()), "TB/s": t}), flush=True)
    if min(x[0] for x in seen) < 6.10 and max(x[0] for x in seen) > 6.30: break
seen.sort(key=lambda x: x[0])
slow, fast = seen[0][1], (seen[-1][1] if len(seen) > 1 else None)   # the slowest and the fastest placement met (on some boxes every placement is slow)
held = [x[1] for x in seen[1:-1]]
del seen
sets = [{}, {"FFHIP_JPEG_VARIANT": "12"}, {"FFHIP_JPEG_VARIANT": "11"}, {"FFHIP_JPEG_VARIANT": "23"}, {"FFHIP_JPEG_VARIANT": "22"}, {"FFHIP_JPEG_NO_XCD_REMAP": "1"},
        {"FFHIP_JPEG_XCD_CHUNK_LOG2": "4"}, {"FFHIP_JPEG_XCD_CHUNK_LOG2": "9"}]
for s in os.environ.get("SWITCHES", "").split(";"):
    if s: sets.append(dict(kv.split("=") for kv in s.split(",")))
for name, buf in (("slow", slow), ("fast", fast)):
    if buf is None:
        print(json.dumps({"placement": name, "found": False})); continue
    for sw in sets:
        for k, v in sw.items(): capi.setenv(k, v)
        row = {"placement": name, "switches": sw, "TB/s": timed(buf, W * 4), "TB/s_pitch+1K": timed(buf, W * 4 + 1024), "TB/s_pitch+8K": timed(buf, W * 4 + 8192)}
        PATTERN = True
        row["pattern_TB/s"] = timed(buf, W * 4)
        PATTERN = False
        for k in sw: capi.setenv(k, None)
        print(json.dumps(row), flush=True)
# ---- where in the buffer is the slow mode?  Sub-batches of 32 images (an eighth of the output each), all XCDs on each
for name, buf in (("slow", slow), ("fast", fast)):
    if buf is None: continue
    pitch = W * 4; stride = pitch * H
    res = []
    for part in range(8):
        i0 = part * 32
        def step():
            ops.jpeg_recon_batch(geom, 32, ty.data_ptr() + i0 * mcus * 512, tu.data_ptr() + i0 * mcus * 128, tv.data_ptr() + i0 * mcus * 128, q.data_ptr(), 0,
                                 buf.data_ptr() + i0 * stride, pitch, stride, None, 0, st)
        for _ in range(2): step()
        L.ffhip_event_record(e0, st)
        for _ in range(6): step()
        L.ffhip_event_record(e1, st); capi.check(L.ffhip_stream_sync(st))
        res.append(round(7 * 32 * W * H / (L.ffhip_event_elapsed_ms(e0, e1) / 6) / 1e9, 3))
    print(json.dumps({"placement": name, "TB/s_per_eighth_of_the_buffer": res}), flush=True)
