#!/usr/bin/env python3
"""Headline kernel time with its buffers from hipExtMallocWithFlags(hipDeviceMallocContiguous) against plain hipMalloc,
alternating in one process.  Diagnostic for the placement-dependent spread."""
import os, sys, json, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
n = int(os.environ.get("FFHIP_BENCH_IMAGES", "256"))
cols, rows = 240, 135
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
mcus = cols * rows
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
by, bc, bo = n * mcus * 512, n * mcus * 128, n * W * 4 * H
src = torch.randint(-30, 31, (by // 2,), device=dev, dtype=torch.int16)
torch.cuda.synchronize()
def alloc(nbytes, flag):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag) if flag else hip.hipMalloc(C.byref(p), nbytes)
    if rc != 0: raise RuntimeError(f"alloc {nbytes} flag {flag}: hip error {rc}")
    return p.value
order = os.environ.get("ORDER", "0404040404")
for trial, ch in enumerate(order):
    flag = int(ch)
    dummy = alloc(1 + trial * 37_000_003, 0)
    try:
        py, pu, pv, po = alloc(by, flag), alloc(bc, flag), alloc(bc, flag), alloc(bo, flag)
    except RuntimeError as e:
        print(json.dumps({"trial": trial, "flag": flag, "error": str(e)}), flush=True); hip.hipFree(dummy); continue
    hip.hipMemcpy(py, src.data_ptr(), by, 3); hip.hipMemcpy(pu, src.data_ptr(), bc, 3); hip.hipMemcpy(pv, src.data_ptr() + bc, bc, 3)
    def step():
        ops.jpeg_recon_batch(geom, n, py, pu, pv, q.data_ptr(), 0, po, W * 4, W * 4 * H, None, 0, st)
    for _ in range(3): step()
    ts = []
    for _ in range(8):
        L.ffhip_event_record(e0, st); step(); L.ffhip_event_record(e1, st)
        ts.append(L.ffhip_event_elapsed_ms(e0, e1))
    print(json.dumps({"trial": trial, "flag": flag, "min_ms": round(min(ts), 4), "mean_ms": round(sum(ts) / len(ts), 4),
                      "TB/s_mean": round(7 * n * W * H / (sum(ts) / len(ts)) / 1e9, 3), "out": hex(po)}), flush=True)
    torch.cuda.synchronize()
    for p in (py, pu, pv, po, dummy): hip.hipFree(p)
