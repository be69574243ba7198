#!/usr/bin/env python3
"""Where the wave-time of k_hevc_intra_groups goes on a grid of 135-tile pictures (PICTURES, default 8): per-TU timestamps of the
diagnostics build (`make -C ffpic_amd/csrc trace`, wall_clock64 = 100 MHz) added up per wave:
  wait      dependency wait of a TU (flags of other groups)          body   the TU itself, by size and kind
  in-group  between the end of a TU and the start of the next one     start  ticket taken -> first TU of the group
  between   end of a group's last TU -> next ticket                   tail   the wave's last TU -> end of the kernel
Usage: PICTURES=8 python3 tests/tools/diag_intra_trace_grid.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import ctypes as C
from ffpic_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", "libffpic_hip_trace.so")
import bench
dev = torch.device("cuda", 0)
L = capi.require_device(0)
L.ffhip_debug_intra_trace.argtypes = [C.c_void_p]
L.ffhip_debug_intra_trace.restype = None
st = torch.cuda.current_stream().cuda_stream
npic = int(os.environ.get("PICTURES", "8"))
tile, tiles_xy = 512, (15, 9)
t0, _ = synth.hevc_intra_tus(tile, tile, seed=5, tu_mix="c5")
px_, py_ = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}[npic]
gx, gy = tiles_xy[0] * px_, tiles_xy[1] * py_
W, H = gx * tile, gy * tile
tus = np.tile(t0, gx * gy)
k = np.repeat(np.arange(gx * gy), len(t0))
sc = np.where(tus["cidx"] == 0, tile, tile // 2)
tus["x"] = (tus["x"].astype(np.int64) + (k % gx) * sc).astype(np.uint16)
tus["y"] = (tus["y"].astype(np.int64) + (k // gx) * sc).astype(np.uint16)
tus, groups, total = bench.hevc_chain_inputs(W, H, seed=40 + npic, tus=tus)
n = len(tus)
dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev)
dr = torch.randint(-40, 40, (total + 64,), dtype=torch.int16, device=dev)
py = torch.zeros((H, W), dtype=torch.int16, device=dev); pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev); pv = torch.zeros_like(pu)
trace = torch.zeros(13 * n + 16, dtype=torch.int64, device=dev)


def run():
    capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), n, dr.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st))


run(); capi.check(L.ffhip_stream_sync(st))
L.ffhip_debug_intra_trace(trace.data_ptr())
run(); capi.check(L.ffhip_stream_sync(st))
tr = trace.cpu().numpy()
rec = tr[:12 * n].reshape(n, 12)
tick = tr[12 * n:13 * n]
t_begin, t_start, t_end, meta = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
assert (t_end > 0).all(), "some TU left no trace"
tkind = ((meta >> 56) & 7).astype(np.int64); lgk = ((meta >> 60) & 7).astype(np.int64)
ticket = ((meta >> 32) & 0xffffff).astype(np.int64); wave = ((meta >> 12) & 0xfffff).astype(np.int64); kslot = (meta & 0xfff).astype(np.int64)
T0 = int(tick[tick > 0].min()); T1 = int(t_end.max())
span = (T1 - T0) / 100.0
waves = np.unique(wave)
print(f"{npic} pictures: {n} TUs, {int(ticket.max()) + 1} groups, {len(waves)} waves used, kernel span {span:.1f} us")
order = np.lexsort((kslot, ticket))           # by group, then slot
tb, ts, te, tk, wv, kd, lg = t_begin[order], t_start[order], t_end[order], ticket[order], wave[order], tkind[order], lgk[order]
first = np.r_[True, tk[1:] != tk[:-1]]
last = np.r_[tk[1:] != tk[:-1], True]
wait = (ts - tb).sum() / 100.0
body = (te - ts).sum() / 100.0
ingroup = (tb[1:] - te[:-1])[~first[1:]].sum() / 100.0
start = (tb[first] - tick[tk[first]]).sum() / 100.0
# between groups, per wave: sort the groups of a wave by ticket time
g_wave, g_tick, g_end = wv[first], tick[tk[first]], te[last]
o2 = np.lexsort((g_tick, g_wave))
gw, gt, ge = g_wave[o2], g_tick[o2], g_end[o2]
samew = gw[1:] == gw[:-1]
between = (gt[1:] - ge[:-1])[samew].sum() / 100.0
lastg = np.r_[~samew, True]
firstg = np.r_[True, ~samew]
tail = (T1 - ge[lastg]).sum() / 100.0
head = (gt[firstg] - T0).sum() / 100.0
total_wt = span * len(waves)
print(f"wave-time {total_wt / 1e3:.1f} ms over {len(waves)} waves:")
for name, v in (("body", body), ("wait", wait), ("in-group", ingroup), ("start", start), ("between", between), ("head", head), ("tail", tail)):
    print(f"  {name:9s} {v / 1e3:9.2f} ms  {100.0 * v / total_wt:5.1f} %")
print(f"  accounted {100.0 * (body + wait + ingroup + start + between + head + tail) / total_wt:.1f} %")
for kk in sorted(set(zip(lg.tolist(), kd.tolist()))):
    sel = (lg == kk[0]) & (kd == kk[1])
    print(f"  TU {1 << kk[0]:2d}x{1 << kk[0]:<2d} kind {kk[1]}: {int(sel.sum()):8d}  body {((te - ts)[sel]).mean() / 100.0:.2f} us  wait {((ts - tb)[sel]).mean() / 100.0:.2f} us  (waiters: {int(((ts - tb)[sel] > 30).sum())})")
# how busy the machine is over time: TUs in their body per 50-us slice
edges = np.arange(T0, T1 + 5000, 5000)
busy = np.zeros(len(edges) - 1)
for a_, b_ in ((ts, te),):
    lo = np.clip((a_ - T0) // 5000, 0, len(busy) - 1).astype(np.int64); hi = np.clip((b_ - T0) // 5000, 0, len(busy) - 1).astype(np.int64)
    np.add.at(busy, lo, 1.0)   # (coarse: a TU counted in the slice it starts in)
print("  TUs started per 50 us slice: " + " ".join(str(int(v)) for v in busy))
