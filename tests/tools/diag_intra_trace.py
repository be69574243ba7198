#!/usr/bin/env python3
"""Where the time of k_hevc_intra_groups goes on one 8K picture: per-TU timestamps (diagnostics build of the library,
`make -C ffpic_amd/csrc trace`, wall_clock64 = 100 MHz) and the critical path reconstructed from them on the host.

For the TU that finishes last, walk backwards: the predecessor of a TU is whichever finished last among the previous TU
of its group and the TUs of other groups whose samples it reads.  Every step of the walk is classified:
  body      t_end - t_start of the TU itself, by TU size
  in-group  t_start(TU) - t_end(previous TU of the group)          (slot decode, prefetch issue)
  hand-off  t_start(TU) - t_end(TU of another group it waited for) (store drain, flag, poll, first gather)
  start     group start: ticket taken -> first TU started, when nothing else explains the start
Usage: diag_intra_trace.py [quadtree|c5] [window_log2]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, synth

capi.LIB_PATH = os.path.join(ROOT, "ffpic_amd", "libffpic_hip_trace.so")
import ctypes as C
dev = torch.device("cuda", 0)
L = capi.require_device(0)
L.ffhip_debug_intra_trace.argtypes = [C.c_void_p]
L.ffhip_debug_intra_trace.restype = None
st = torch.cuda.current_stream().cuda_stream
which = sys.argv[1] if len(sys.argv) > 1 else "quadtree"
if len(sys.argv) > 2:
    capi.setenv("FFHIP_HEVC_INTRA_WINDOW", sys.argv[2])
W, H = (7680, 4352) if not os.environ.get("TRACE_SMALL") else (1920, 1088)
tus, res = synth.hevc_intra_tus(W, H, seed=2 if which == "quadtree" else 5, tu_mix=None if which == "quadtree" else "c5")
n = len(tus)
dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev); dr = torch.from_numpy(res).to(dev)
py = torch.zeros((H, W), dtype=torch.int16, device=dev); pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev); pv = torch.zeros_like(pu)
trace = torch.zeros(13 * n + 16, dtype=torch.int64, device=dev)


def run():
    capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), n, dr.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st))


run(); capi.check(L.ffhip_stream_sync(st))
L.ffhip_debug_intra_trace(trace.data_ptr())
run(); capi.check(L.ffhip_stream_sync(st))
tr = trace.cpu().numpy()
rec = tr[:12 * n].reshape(n, 12)
tick = tr[12 * n:]
t_begin, t_start, t_end, meta = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
assert (t_end > 0).all(), "some TU left no trace"
tkind = ((meta >> 56) & 7).astype(np.int64); halo = ((meta >> 59) & 1).astype(np.int64)
ticket = ((meta >> 32) & 0xffffff).astype(np.int64); wave = ((meta >> 12) & 0xfffff).astype(np.int64); kslot = (meta & 0xfff).astype(np.int64)
T0 = tick[tick > 0].min()
us = lambda t: (t - T0) / 100.0
print(f"{which}: {n} TUs, {int(ticket.max()) + 1} groups, {len(np.unique(wave))} waves used, kernel span {us(t_end.max()):.1f} us")

# owner maps (4x4 blocks) per plane and who precedes whom inside a group
lg = tus["log2_size"].astype(np.int64); cidx = tus["cidx"].astype(np.int64)
X = tus["x"].astype(np.int64); Y = tus["y"].astype(np.int64)
planes = [(W, H), (W // 2, H // 2), (W // 2, H // 2)]
owner = [np.full((ph // 4, pw // 4), -1, np.int64) for (pw, ph) in planes]
for c in range(3):
    for l in range(2, 6):
        sel = np.nonzero((cidx == c) & (lg == l))[0]
        b = 1 << (l - 2)
        for by in range(b):
            for bx in range(b):
                owner[c][Y[sel] // 4 + by, X[sel] // 4 + bx] = sel
order = np.lexsort((kslot, ticket))
prev_in_group = np.full(n, -1, np.int64)
same = ticket[order][1:] == ticket[order][:-1]
prev_in_group[order[1:][same]] = order[:-1][same]
at = tus["avail_top"].astype(np.uint64); al = tus["avail_left"].astype(np.uint64); fl = tus["flags"].astype(np.int64)


def cross_deps(i):
    c, nn, x0, y0 = int(cidx[i]), 1 << int(lg[i]), int(X[i]), int(Y[i])
    out = set()
    own = owner[c]
    def dep(px, py):
        j = int(own[py // 4, px // 4])
        if 0 <= j < i and ticket[j] != ticket[i]:
            out.add(j)
    if fl[i] & 1: dep(x0 - 1, y0 - 1)
    for k in range(0, 2 * nn, 4):
        if (int(at[i]) >> k) & 0xf: dep(x0 + k, y0 - 1)
        if (int(al[i]) >> k) & 0xf: dep(x0 - 1, y0 + k)
    return out


stage_names = ["gather", "filter", "residual", "predict+store", "publish"]
stage = {l: np.zeros(5) for l in (2, 3, 4, 5)}
stage_n = {l: 0 for l in (2, 3, 4, 5)}
pk = {}   # program TUs on the path: (size, kind, halo) -> [count, body ticks]
cat = {"body": {2: 0, 3: 0, 4: 0, 5: 0}, "in-group": 0, "hand-off": 0, "start": 0}
cnt = {"body": {2: 0, 3: 0, 4: 0, 5: 0}, "in-group": 0, "hand-off": 0, "start": 0}
handoffs = []
gaps = {}
cur = int(np.argmax(t_end))
steps = 0
while cur >= 0:
    steps += 1
    cat["body"][int(lg[cur])] += int(t_end[cur] - t_start[cur]); cnt["body"][int(lg[cur])] += 1
    if tkind[cur]:
        e = pk.setdefault((int(lg[cur]), int(tkind[cur]), int(halo[cur])), [0, 0]); e[0] += 1; e[1] += int(t_end[cur] - t_start[cur])
    if tkind[cur] == 0:   # the stage stamps belong to the generic body
        st4 = rec[cur, 4:8]
        stage[int(lg[cur])] += np.array([st4[0] - t_start[cur], st4[1] - st4[0], st4[2] - st4[1], st4[3] - st4[2], t_end[cur] - st4[3]], dtype=np.float64)
        stage_n[int(lg[cur])] += 1
    cands = []
    p = int(prev_in_group[cur])
    if p >= 0: cands.append((int(t_end[p]), "in-group", p))
    for j in cross_deps(cur): cands.append((int(t_end[j]), "hand-off", j))
    if not cands or (p < 0 and max(cands)[0] < int(tick[ticket[cur]])):
        cat["start"] += int(t_start[cur] - tick[ticket[cur]]); cnt["start"] += 1
        if not cands: break
        # the group was picked up after everything it needed was there: continue from the ticket holder's previous group
        same_wave = np.nonzero((wave == wave[cur]) & (t_end <= tick[ticket[cur]]))[0]
        if len(same_wave) == 0: break
        cur = int(same_wave[np.argmax(t_end[same_wave])])
        continue
    te, kind, j = max(cands)
    gap = int(t_start[cur]) - te
    cat[kind] += gap; cnt[kind] += 1
    if kind == "in-group" and tkind[cur] == 0 and tkind[j] == 0:   # generic after generic: the pieces of the gap (stamps 4..6 sit in the later TU's record)
        g4, g5, g6, g7 = int(rec[cur, 8]), int(rec[cur, 9]), int(rec[cur, 10]), int(rec[cur, 11])
        if g5 >= te and g6 <= int(t_start[cur]):
            gg = gaps.setdefault("generic->generic", [0, 0, 0, 0, 0, 0, 0, 0]); gg[0] += 1
            gg[2] += g5 - te; gg[3] += g6 - g5; gg[4] += int(t_begin[cur]) - g6; gg[5] += int(t_start[cur]) - int(t_begin[cur])
    if kind == "hand-off": handoffs.append(gap)
    cur = j
tot = sum(cat["body"].values()) + cat["in-group"] + cat["hand-off"] + cat["start"]
print(f"critical path: {steps} TUs, {tot / 100.0:.1f} us accounted")
for l in (2, 3, 4, 5):
    if cnt["body"][l]:
        print(f"  body {1 << l:2d}x{1 << l:<2d}: {cnt['body'][l]:6d} TUs  {cat['body'][l] / 100.0:9.1f} us  ({cat['body'][l] / cnt['body'][l] / 100.0:.2f} us each)")
for l in (2, 3, 4, 5):
    if stage_n[l]:
        print(f"    generic {1 << l:2d}x{1 << l:<2d} ({stage_n[l]} TUs), stages (us each): " + ", ".join(f"{nm} {stage[l][q] / stage_n[l] / 100.0:.2f}" for q, nm in enumerate(stage_names)))
for key in sorted(pk):
    print(f"    program {1 << key[0]}x{1 << key[0]} kind {key[1]} halo {key[2]}: {pk[key][0]:6d} TUs, {pk[key][1] / pk[key][0] / 100.0:.2f} us each")
print(f"  programs overall: {int((tkind > 0).sum())} of {n} TUs")
for k in ("in-group", "hand-off", "start"):
    if cnt[k]:
        print(f"  {k:9s}: {cnt[k]:6d} steps {cat[k] / 100.0:9.1f} us  ({cat[k] / cnt[k] / 100.0:.2f} us each)")
for kname, gg in gaps.items():
    c = gg[0]
    print(f"  gap {kname}: {c} steps; trace write + sync + program fetch + swap {gg[2] / c / 100:.2f}, extras into place {gg[3] / c / 100:.2f}, loop top {gg[4] / c / 100:.2f}, wait {gg[5] / c / 100:.2f} us")
if handoffs:
    h = np.array(handoffs) / 100.0
    print(f"  hand-off gap percentiles (us): p10 {np.percentile(h, 10):.2f} p50 {np.percentile(h, 50):.2f} p90 {np.percentile(h, 90):.2f} max {h.max():.2f}")
# whole-kernel accounting per wave
body = (t_end - t_start).sum() / 100.0; wait = (t_start - t_begin).sum() / 100.0
print(f"all waves: body {body / 1e3:.1f} ms, waiting at flags {wait / 1e3:.1f} ms, over {len(np.unique(wave))} waves x {us(t_end.max()) / 1e3:.2f} ms")
# how far ahead of need do tickets go out: first-TU wait per group
first = np.nonzero(kslot == 0)[0]
w0 = (t_start[first] - t_begin[first]) / 100.0
print(f"first TU of a group waits: mean {w0.mean():.1f} us, p50 {np.percentile(w0, 50):.1f}, p90 {np.percentile(w0, 90):.1f}; slot-ready after ticket: mean {((t_begin[first] - tick[ticket[first]]) / 100.0).mean():.2f} us")
