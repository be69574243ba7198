#!/usr/bin/env python3
"""Per-stage timings of the non-headline kernels at BASELINE config 4/5 sizes (HIP events on the
launch stream, inputs resident in HBM).  Diagnostic: numbers quoted in DESIGN.md section 4."""
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

def timeit(fn, reps=10):
    for _ in range(3): fn()
    L.ffhip_event_record(e0, st)
    for _ in range(reps): fn()
    L.ffhip_event_record(e1, st)
    return L.ffhip_event_elapsed_ms(e0, e1) / reps

out = {}
# --- planar colour, 8K 4:2:0 (C5) and 64 x 1080p (C4-like batch)
for tag, (H, W, n, sixteen) in {"yuv420_8bit_64x1080p": (1088, 1920, 64, False), "yuv420_16bit_8K": (4352, 7680, 1, True)}.items():
    dt = torch.int16 if sixteen else torch.uint8
    y = torch.randint(0, 256, (n, H, W), device=dev).to(dt); u = torch.randint(0, 256, (n, H // 2, W // 2), device=dev).to(dt); v = u.clone()
    o = torch.empty((n, H, W * 4), dtype=torch.uint8, device=dev)
    if sixteen:
        f = lambda: capi.check(L.ffhip_yuv420_to_bgra_16(o.data_ptr(), W * 4, y.data_ptr(), u.data_ptr(), v.data_ptr(), W, W // 2, H // 64, W // 64, 64, n, H * W, H * W // 4, H * W * 4, st))
        bpp = 3 + 4
    else:
        f = lambda: capi.check(L.ffhip_yuv420_to_bgra(o.data_ptr(), W * 4, y.data_ptr(), u.data_ptr(), v.data_ptr(), W, W // 2, H // 16, W // 16, n, H * W, H * W // 4, H * W * 4, st))
        bpp = 1.5 + 4
    ms = timeit(f)
    out[tag] = {"ms": round(ms, 4), "Gpx/s": round(n * H * W / ms / 1e6, 1), "GB/s": round(bpp * n * H * W / ms / 1e6, 1)}
# --- VP8 residual: 64 frames of 8160 MBs
n_mb = 8160 * 64
lv, info = synth.vp8_macroblocks(8160, seed=1)
tl = torch.from_numpy(lv).to(dev).repeat(64, 1, 1); ti = torch.from_numpy(info).to(dev).repeat(64, 1)
tq = torch.from_numpy(synth.vp8_quant().astype(np.int16)).to(dev)
tr = torch.empty((n_mb, 384), dtype=torch.int16, device=dev)
ms = timeit(lambda: capi.check(L.ffhip_vp8_residual_batch(n_mb, tl.data_ptr(), ti.data_ptr(), tq.data_ptr(), tr.data_ptr(), st)))
out["vp8_residual_64x1080p"] = {"ms": round(ms, 4), "Gpx/s": round(n_mb * 256 / ms / 1e6, 1), "GB/s": round(n_mb * (800 + 32 + 768) / ms / 1e6, 1)}
# --- HEVC residual at 8K: 32x32 and 16x16 TUs covering 7680x4320 luma
for n, cnt in ((32, 4 * 240 * 135), (16, 4 * 480 * 270), (8, 4 * 960 * 540), (4, 4 * 1920 * 1080)):   # four 8K luma planes per launch
    lvl = torch.randint(-20, 21, (cnt, n * n), device=dev).to(torch.int16)
    info = torch.zeros((cnt, 4), dtype=torch.uint8, device=dev); info[:, 0] = 27
    res = torch.empty_like(lvl)
    ms = timeit(lambda: capi.check(L.ffhip_hevc_residual_batch(n, cnt, lvl.data_ptr(), info.data_ptr(), None, 8, 0, res.data_ptr(), st)))
    out[f"hevc_residual_{n}x{n}_4x8K_luma"] = {"ms": round(ms, 4), "Gsamples/s": round(cnt * n * n / ms / 1e6, 1), "GB/s": round(4 * cnt * n * n / ms / 1e6, 1)}
# --- VP8 predict + recon, 16 frames of 1080p
c, r, nf = 120, 68, 16
modes = np.stack([synth.vp8_modes(c, r, seed=i) for i in range(nf)])
resid = torch.from_numpy(np.stack([synth.vp8_residual(c * r, seed=i) for i in range(nf)])).to(dev)
dm = torch.from_numpy(modes).to(dev)
Y = torch.zeros((nf, 16 * r, 16 * c), dtype=torch.uint8, device=dev); U = torch.zeros((nf, 8 * r, 8 * c), dtype=torch.uint8, device=dev); V = torch.zeros_like(U)
import time
def pred():
    capi.check(L.ffhip_vp8_predict_recon(c, r, nf, modes.ctypes.data, dm.data_ptr(), resid.data_ptr(), c * r * 384, None, Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * c * r, 64 * c * r, st))
for name, env in (("levels", {"FFHIP_VP8_PRED_MODE": "levels"}), ("rows", {})):
    capi.setenv("FFHIP_VP8_PRED_MODE", None)
    [capi.setenv(k_, v_) for k_, v_ in env.items()]
    pred(); capi.check(L.ffhip_stream_sync(st))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); pred(); capi.check(L.ffhip_stream_sync(st)); best = min(best, (time.perf_counter() - t0) * 1e3)
    out[f"vp8_predict_recon_16x1080p_{name}"] = {"wall_ms": round(best, 3), "Mpx/s": round(nf * 256 * c * r / best / 1e3, 1)}
capi.setenv("FFHIP_VP8_PRED_MODE", None)
# --- VP8 loop filter, same 16 frames
flt = torch.from_numpy(synth.vp8_filters(seed=3)).to(dev)
def lf():
    capi.check(L.ffhip_vp8_loopfilter(c, r, nf, 2, dm.data_ptr(), flt.data_ptr(), Y.data_ptr(), U.data_ptr(), V.data_ptr(), 256 * c * r, 64 * c * r, st))
for name, env in (("levels", {"FFHIP_VP8_LF_MODE": "levels"}), ("rows", {})):
    capi.setenv("FFHIP_VP8_LF_MODE", None)
    [capi.setenv(k_, v_) for k_, v_ in env.items()]
    lf(); capi.check(L.ffhip_stream_sync(st))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); lf(); capi.check(L.ffhip_stream_sync(st)); best = min(best, (time.perf_counter() - t0) * 1e3)
    out[f"vp8_loopfilter_normal_16x1080p_{name}"] = {"wall_ms": round(best, 3), "Mpx/s": round(nf * 256 * c * r / best / 1e3, 1)}
capi.setenv("FFHIP_VP8_LF_MODE", None)
# --- HEVC intra recon: one 1920x1088 picture, then the 8K picture of config 5, level launches vs grouped single launch
def intra_case(tag, W, H, seed, envs):
    tus, res = synth.hevc_intra_tus(W, H, seed=seed)
    dtus = torch.from_numpy(tus.view(np.uint8).copy()).to(dev); dres = torch.from_numpy(res).to(dev)
    py = torch.zeros((H, W), dtype=torch.int16, device=dev); pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev); pv = torch.zeros_like(pu)
    def intra():
        capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dtus.data_ptr(), len(tus), dres.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st))
    for name, env in envs:
        for k in ("FFHIP_HEVC_INTRA_MODE", "FFHIP_HEVC_INTRA_WINDOW", "FFHIP_HEVC_INTRA_WAVES", "FFHIP_HEVC_INTRA_DECODE_ORDER"):
            capi.setenv(k, None)
        [capi.setenv(k_, v_) for k_, v_ in env.items()]
        intra(); capi.check(L.ffhip_stream_sync(st))
        best_wall, best_dev = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            intra()
            capi.check(L.ffhip_stream_sync(st))
            best_wall = min(best_wall, (time.perf_counter() - t0) * 1e3)
        out[f"hevc_intra_recon_{tag}_{name}"] = {"wall_ms": round(best_wall, 3), "tus": int(len(tus)), "Mpx/s": round(W * H / best_wall / 1e3, 1)}
    for k in ("FFHIP_HEVC_INTRA_MODE", "FFHIP_HEVC_INTRA_WINDOW"):
        capi.setenv(k, None)
envs = [("levels", {"FFHIP_HEVC_INTRA_MODE": "levels"})] + [(f"groups_w{1 << w}", {"FFHIP_HEVC_INTRA_WINDOW": str(w)}) for w in (6, 5, 4, 3)]
if "--sweep" in sys.argv:
    envs += [(f"w32_waves{n}", {"FFHIP_HEVC_INTRA_WAVES": str(n)}) for n in (128, 256, 512, 2048, 4096)]
    envs += [(f"w16_waves{n}", {"FFHIP_HEVC_INTRA_WAVES": str(n), "FFHIP_HEVC_INTRA_WINDOW": "4"}) for n in (512, 2048, 4096)]
    envs += [("w32_decode_order_4096", {"FFHIP_HEVC_INTRA_WAVES": "4096", "FFHIP_HEVC_INTRA_DECODE_ORDER": "1"})]
intra_case("1080p", 1920, 1088 + 64 - 1088 % 64 if 1088 % 64 else 1088, 1, envs)
if "--8k" in sys.argv:
    intra_case("8K", 7680, 4352, 2, envs)
print(json.dumps(out, indent=1))
