#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the fused JPEG reconstruction (dequant + IDCT + YUV->BGRA)
on MI355X, the metric BASELINE.json names.

  python bench.py --gpus N --steps K --warmup W       (N > 1 without a launcher: spawns its own N ranks as a child torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: 256 synthetic 3840x2160 4:2:0 coefficient grids
(BASELINE config 3) already resident in HBM, reconstructed to BGRA in HBM by ONE launch of
k_jpeg420_fused per GPU.  Images are independent, so ranks own disjoint contiguous image ranges
(ffhip_shard_range) and the only collective is the batch close: one RCCL all-gather of a 32-byte
{rank, status, first, count, checksum} record per rank, issued from C (ffhip_batch_close).
Default scaling is WEAK -- the path partitions into independent images with no data-path collective, so every GPU takes the
configuration's 256 images (at N = 1 that IS BASELINE config 3; at N = 8 it is 2 048 images, 256 per GPU);
`--scaling strong` is config 3 read literally: 256 images in total, 32 per GPU at N = 8.  Rank 0 prints one JSON line.

At N = 1 the line also carries `extra`: BASELINE configs 2, 4 and 5 measured the same way (HIP events on the
launch stream, inputs resident in HBM), each with its own roofline figures, parity flag and CPU baseline.

PyTorch is plumbing here: device memory, the stream handle and torch.distributed for process bootstrap.
The compute goes through the C ABI of ffpic_amd/libffpic_hip.so.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _self_launch():
    """`python3 bench.py --gpus N` with N > 1 and no launcher environment: this process -- which has imported neither torch
    nor the HIP library, i.e. has made no GPU call -- starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD (never an exec), relays the one JSON line rank 0 prints and exits with the child's status."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    gpus = ap.parse_known_args()[0].gpus
    if gpus <= 1:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL between processes needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    for line in child.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    sys.exit(child.wait())


if __name__ == "__main__":
    _self_launch()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ffpic_amd import capi, ops, shard, synth  # noqa: E402

WORKLOADS = {
    # name: (mcu_cols, mcu_rows, images in the batch, description)
    "c3": (240, 135, 256, "C3: 256 x 3840x2160 4:2:0 JPEG coefficient grids"),
    "c2": (120, 68, 1024, "C2: 1024 x 1920x1088 4:2:0 JPEG coefficient grids"),
}
BYTES_PER_PIXEL = 7.0   # 3 B int16 coefficients (1.5 samples) + 4 B BGRA at 4:2:0
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def gen_batch_on_device(dev, cols, rows, n_images, seed):
    """Synthetic coefficient planes of SURVEY.md 8d generated directly in HBM (the numpy
    generator of ffpic_amd.synth draws from the same distribution; this one uses the
    device RNG so that 6 GB of input does not cross PCIe)."""
    q = synth.quant_tables()
    rank = torch.from_numpy(synth.zigzag_rank()).to(dev)
    scale = 8.0 * torch.exp(-rank.double() / 6.0).float()
    gen = torch.Generator(device=dev)
    gen.manual_seed(synth.SEED_BASE + seed)
    mcus = cols * rows

    def plane(blocks_per_image, qrow):
        lim = torch.from_numpy((2047 // q[qrow].astype(np.int64)).astype(np.float32)).to(dev)
        out = torch.empty((n_images * blocks_per_image, 64), dtype=torch.int16, device=dev)
        chunk = max(1, (1 << 25) // (blocks_per_image * 64))
        for i in range(0, n_images, chunk):
            nb = min(chunk, n_images - i) * blocks_per_image
            u = torch.rand((nb, 64), generator=gen, device=dev) - 0.5
            lap = -torch.sign(u) * torch.log1p(-2.0 * u.abs().clamp(max=0.4999999))
            c = torch.round(lap * scale)
            c[:, 0] = torch.round(torch.randn((nb,), generator=gen, device=dev) * 60.0)
            c = torch.minimum(torch.maximum(c, -lim), lim)
            out[i * blocks_per_image: i * blocks_per_image + nb] = c.to(torch.int16)
        return out.view(-1)

    return plane(mcus * 4, 0), plane(mcus, 1), plane(mcus, 1), q


def oracle_lib():
    """The checker (tests/oracle_lib.py): used only for parity spot checks and the cpu_baseline legs, never timed as
    the product and never on its path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


def cpu_baseline(cols, rows, t_y, t_u, t_v, q, budget_s=20.0):
    """The reference's own C (oracle/_ref, compiled from /root/reference in the build
    container) -- or our bit-exact port when that .so did not travel -- timed on the
    host cores over a bounded sample of the same workload."""
    O = oracle_lib()
    from concurrent.futures import ThreadPoolExecutor
    mcus = cols * rows
    cores = max(1, min(os.cpu_count() or 1, 16, t_u.numel() // (mcus * 64)))
    g = O.make_geom(cols, rows)
    use_ref = os.path.exists(O.REF_SO)
    kind = "reference" if use_ref else "port"
    H, W = g.height, g.width

    def one(i, cy, cu, cv):
        if use_ref:
            O.ref_jpeg_recon(g, cy, cu, cv, q)
        else:
            O.oracle_jpeg_recon(g, cy, cu, cv, q)

    imgs = []
    for i in range(cores):
        imgs.append((t_y[i * mcus * 256:(i + 1) * mcus * 256].cpu().numpy(),
                     t_u[i * mcus * 64:(i + 1) * mcus * 64].cpu().numpy(),
                     t_v[i * mcus * 64:(i + 1) * mcus * 64].cpu().numpy()))
    t0 = time.perf_counter()
    one(0, *imgs[0])                       # size the sample from one image on one core
    t1 = time.perf_counter() - t0
    per_core = max(1, min(8, int(budget_s / max(t1, 1e-3) / cores)))
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda k: [one(k, *imgs[k]) for _ in range(per_core)], range(cores)))
        dt = time.perf_counter() - t0
    n = cores * per_core
    return {"value": round(n * H * W / dt / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": kind,
            "sample": f"{n} images of {W}x{H} ({per_core} per thread, {cores} threads, ctypes releases the GIL); "
                      f"1 thread: {H * W / t1 / 1e6:.1f} Mpixels/s"}


def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), already
    corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, KiB units).  A constant read from
    profiles/latest_pmc.json, not a counter collected in this run (counters need their own rocprofv3 pass)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        if rec.get("workload") == workload:
            return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


class Timer:
    """HIP events on the launch stream (ffhip_event_*): average duration of `fn` over `reps` back-to-back calls."""

    def __init__(self, L, stream):
        self.L, self.stream = L, stream
        self.e0, self.e1 = L.ffhip_event_create(), L.ffhip_event_create()

    def ms(self, fn, reps=10, warm=2):
        """`warm` untimed calls, then `reps` timed ones -- twice, the second block counting (FFHIP_BENCH_PASSES=1: once, as until round 6).  The extras run one
        after the other with host work (data generation, oracle checks) in between, during which the GPU's clocks fall back; a kernel bound by its arithmetic
        measured in the first five milliseconds behind such a pause reads 10-15 % low (k_vp8_residual: 0.69 of the HBM peak in a first block of ten launches, 0.74-0.78
        in the second), an access pattern hardly at all."""
        for _ in range(warm):
            fn()
        t = None
        for _ in range(1 if os.environ.get("FFHIP_BENCH_PASSES") == "1" else 2):
            capi.check(self.L.ffhip_event_record(self.e0, self.stream))
            for _ in range(reps):
                fn()
            capi.check(self.L.ffhip_event_record(self.e1, self.stream))
            capi.check(self.L.ffhip_stream_sync(self.stream))
            t = self.L.ffhip_event_elapsed_ms(self.e0, self.e1) / reps
        return t


def roof(bytes_per_launch, ms):
    gbs = bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_launch": int(bytes_per_launch), "kernel_ms": round(ms, 4)}


# --------------------------------------------------------------------------------------------------------------
# extra: BASELINE configs 2, 4, 5 (N = 1 only, after the headline timing)

def extra_c2(L, dev, stream, T):
    """configs[1]: 1024 x 1920x1088 grids, one launch of k_jpeg420_fused"""
    cols, rows, n = 120, 68, 1024
    geom = capi.jpeg_geom(cols, rows)
    H, W = geom.height, geom.width
    t_y, t_u, t_v, q = gen_batch_on_device(dev, cols, rows, n, seed=77)
    t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
    out = torch.empty(n * W * 4 * H, dtype=torch.uint8, device=dev)

    def step():
        ops.jpeg_recon_batch(geom, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H,
                             None, 0, stream)
    ms = T.ms(step, reps=10, warm=3)
    O = oracle_lib()
    mcus = cols * rows
    parity = True
    for i in (0, n - 1):
        exp = O.oracle_jpeg_recon(O.make_geom(cols, rows), t_y[i * mcus * 256:(i + 1) * mcus * 256].cpu().numpy(),
                                  t_u[i * mcus * 64:(i + 1) * mcus * 64].cpu().numpy(), t_v[i * mcus * 64:(i + 1) * mcus * 64].cpu().numpy(), q)[0]
        parity = parity and bool(np.array_equal(out[i * W * 4 * H:(i + 1) * W * 4 * H].cpu().numpy().reshape(H, W, 4), exp))
    res = {"workload": "C2: 1024 x 1920x1088 4:2:0 JPEG coefficient grids, one launch", "value": round(n * H * W / ms / 1e3, 1), "unit": "Mpixels/s",
           "ms_per_step": round(ms, 4), "parity_vs_oracle_first_and_last_image": parity,
           "roofline": dict(roof(BYTES_PER_PIXEL * n * H * W, ms), kernel="k_jpeg420_fused")}
    del t_y, t_u, t_v, out
    torch.cuda.empty_cache()
    return res


def extra_layouts(L, dev, stream, T):
    """The other baseline sampling layouts (k_jpeg_fused_strip) at the headline's batch size: 256 x 3840x2160 (coded 3840x2176),
    one launch each; algorithmic bytes per pixel 2 B per coefficient sample + 4 B BGRA.  Image 0 against the oracle."""
    n, W, H = 256, 3840, 2176
    q = synth.quant_tables()
    t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
    O = oracle_lib()
    res = {}
    for name, (nc, h, v) in {"444": (3, 1, 1), "422": (3, 2, 1), "440": (3, 1, 2), "grey": (1, 1, 1), "411": (3, 4, 1), "114": (3, 1, 4)}.items():
        cols, rows = W // (8 * h), H // (8 * v)
        g = capi.jpeg_geom(cols, rows, nc, h, v, (0, 1, 1))
        by, bc = cols * rows * h * v, cols * rows
        ty = torch.randint(-30, 31, (n * by, 64), device=dev, dtype=torch.int16)
        tu = torch.randint(-30, 31, (n * bc, 64), device=dev, dtype=torch.int16) if nc == 3 else None
        tv = torch.randint(-30, 31, (n * bc, 64), device=dev, dtype=torch.int16) if nc == 3 else None
        out = torch.empty(n * W * H * 4, dtype=torch.uint8, device=dev)

        def step():
            ops.jpeg_recon_batch(g, n, ty.data_ptr(), tu.data_ptr() if nc == 3 else None, tv.data_ptr() if nc == 3 else None, t_q.data_ptr(), 0,
                                 out.data_ptr(), W * 4, W * 4 * H, None, 0, stream)
        ms = T.ms(step, reps=10, warm=3)
        exp = O.oracle_jpeg_recon(O.make_geom(cols, rows, nc, h, v, (0, 1, 1)), ty[:by].cpu().numpy(), tu[:bc].cpu().numpy() if nc == 3 else None,
                                  tv[:bc].cpu().numpy() if nc == 3 else None, q)[0]
        parity = bool(np.array_equal(out[:W * H * 4].cpu().numpy().reshape(H, W, 4), exp))
        bpp = 4 + 2 * (1 + (2.0 / (h * v) if nc == 3 else 0))
        # the layout's bare access pattern on the same buffers (ffhip_jpeg_pattern_calibrate; `out` holds meaningless bytes afterwards)
        def cal():
            capi.check(L.ffhip_jpeg_pattern_calibrate(C.byref(g), n, ty.data_ptr(), tu.data_ptr() if nc == 3 else None, tv.data_ptr() if nc == 3 else None,
                                                      t_q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H, stream), "ffhip_jpeg_pattern_calibrate")
        try:
            ms_pat = T.ms(cal, reps=10, warm=2)
        except Exception:
            ms_pat = None
        res[name] = {"ms_per_step": round(ms, 4), "value": round(n * W * H / ms / 1e3, 1), "unit": "Mpixels/s", "parity_vs_oracle_first_image": parity,
                     "roofline": dict(roof(bpp * n * W * H, ms), kernel="k_jpeg_fused_strip",
                                      pattern_GBps=None if ms_pat is None else round(bpp * n * W * H / ms_pat / 1e6, 1),
                                      frac_of_pattern=None if ms_pat is None else round(ms_pat / ms, 4))}
        del ty, tu, tv, out
        torch.cuda.empty_cache()
    return {"workload": "256 x 3840x2160 grids of the other baseline layouts (4:4:4, 4:2:2, 4:4:0, grey, 4:1:1 = h4v1 and its transpose h1v4), one launch each", **res}


def extra_stage_kernels(L, dev, stream, T):
    """The HBM-bound stage kernels of configs 4 and 5 alone, at batch sizes that fill the chip (inside the chains above they run
    on one picture's worth of data, i.e. launch-bound): VP8 residual on 256 x 8160 macroblocks, the HEVC residual kernels on
    sixteen 8K luma planes per TU size, planar colour on 256 x 1080p (8 bit) and sixteen 8K pictures (16 bit).  HIP events over ten
    back-to-back launches (the gaps between launches are in: profiles/ has the rocprofv3 averages of the kernels alone)."""
    out = {}
    n_mb = 8160 * 256
    lv, info = synth.vp8_macroblocks(8160, seed=1)
    tl = torch.from_numpy(lv).to(dev).repeat(256, 1, 1); ti = torch.from_numpy(info).to(dev).repeat(256, 1)
    tq = torch.from_numpy(synth.vp8_quant().astype(np.int16)).to(dev)
    tr = torch.empty((n_mb, 384), dtype=torch.int16, device=dev)
    ms = T.ms(lambda: capi.check(L.ffhip_vp8_residual_batch(n_mb, tl.data_ptr(), ti.data_ptr(), tq.data_ptr(), tr.data_ptr(), stream)), reps=10, warm=3)
    out["vp8_residual_256x1080p"] = dict(roof(n_mb * (800 + 32 + 768), ms), kernel="k_vp8_residual")
    # the kernel's loads and stores without its arithmetic, on the same buffers (FFHIP_VP8_RESIDUAL_PATTERN=1: `tr` holds meaningless bytes afterwards)
    try:
        capi.setenv("FFHIP_VP8_RESIDUAL_PATTERN", "1")
        ms_pat = T.ms(lambda: capi.check(L.ffhip_vp8_residual_batch(n_mb, tl.data_ptr(), ti.data_ptr(), tq.data_ptr(), tr.data_ptr(), stream)), reps=10, warm=3)
        out["vp8_residual_256x1080p"]["frac_of_pattern"] = round(ms_pat / ms, 4)
    finally:
        capi.setenv("FFHIP_VP8_RESIDUAL_PATTERN", None)
    del tl, ti, tr
    for n, cnt in ((32, 16 * 240 * 135), (16, 16 * 480 * 270), (8, 16 * 960 * 540), (4, 16 * 1920 * 1080)):
        lvl = torch.randint(-20, 21, (cnt, n * n), device=dev, dtype=torch.int16)
        inf = torch.zeros((cnt, 4), dtype=torch.uint8, device=dev); inf[:, 0] = 27
        res = torch.empty_like(lvl)
        ms = T.ms(lambda: capi.check(L.ffhip_hevc_residual_batch(n, cnt, lvl.data_ptr(), inf.data_ptr(), None, 8, 0, res.data_ptr(), stream)), reps=10, warm=3)
        out[f"hevc_residual_{n}x{n}_16x8K_luma"] = dict(roof(4 * cnt * n * n, ms), kernel="k_hevc_residual" + (f"{n}_mfma" if n > 4 else "4"))
        del lvl, inf, res
    for tag, (H, W, n, sixteen) in {"yuv420_8bit_256x1080p": (1088, 1920, 256, False), "yuv420_16bit_16x8K": (4352, 7680, 16, True)}.items():
        dt = torch.int16 if sixteen else torch.uint8
        y = torch.randint(0, 256, (n, H, W), device=dev, dtype=dt); u = torch.randint(0, 256, (n, H // 2, W // 2), device=dev, dtype=dt); v = u.clone()
        o = torch.empty((n, H, W * 4), dtype=torch.uint8, device=dev)
        if sixteen:
            f = lambda: capi.check(L.ffhip_yuv420_to_bgra_16(o.data_ptr(), W * 4, y.data_ptr(), u.data_ptr(), v.data_ptr(), W, W // 2, H // 64, W // 64, 64, n, H * W, H * W // 4, H * W * 4, stream))
        else:
            f = lambda: capi.check(L.ffhip_yuv420_to_bgra(o.data_ptr(), W * 4, y.data_ptr(), u.data_ptr(), v.data_ptr(), W, W // 2, H // 16, W // 16, n, H * W, H * W // 4, H * W * 4, stream))
        ms = T.ms(f, reps=10, warm=3)
        out[tag] = dict(roof((3 + 4 if sixteen else 1.5 + 4) * n * H * W, ms), kernel="k_yuv420_to_bgra")
        del y, u, v, o
    torch.cuda.empty_cache()
    return {"workload": "the HBM-bound stage kernels alone at chip-filling batch sizes", **out}


def _guarded(shape_rows, stride):
    """A plane with one zeroed, readable row in front (what the reference's 16x16 V_PRED / H_PRED read at the top row)."""
    buf = np.zeros((shape_rows + 1) * stride, np.uint8)
    return buf, buf[stride:]


def c4_cpu_chain_frame(c, r, lv, info, q, modes, ft, filt):
    """ONE key frame through residual -> predict -> loop filter -> BGRA in C, one call (oracle/_ref's ref_vp8_chain_frame: the
    reference's own functions; oracle/ffo_chain.c when the compiled reference did not travel).  Returns (bgra, seconds, kind)."""
    O = oracle_lib()
    use_ref = os.path.exists(O.REF_SO)
    lib = O.ref() if use_ref else O.ffo()
    fn = lib.ref_vp8_chain_frame if use_ref else lib.ffo_vp8_chain_frame
    vp = C.c_void_p
    fn.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int]
    fn.restype = None
    n_mb, Wp, Hp = c * r, 16 * c, 16 * r
    lv, info, q, modes, filt = (np.ascontiguousarray(a) for a in (lv, info, q.astype(np.uint16), modes, filt))
    res = np.zeros((n_mb, 384), np.int16)
    (yb, y), (ub, u), (vb, v) = _guarded(Hp, Wp), _guarded(Hp // 2, Wp // 2), _guarded(Hp // 2, Wp // 2)
    o = np.zeros((Hp, Wp * 4), np.uint8)
    t0 = time.perf_counter()
    fn(c, r, lv.ctypes.data, info.ctypes.data, q.ctypes.data, modes.ctypes.data, ft, filt.ctypes.data, res.ctypes.data, y.ctypes.data, u.ctypes.data,
       v.ctypes.data, o.ctypes.data, Wp * 4)
    return o, time.perf_counter() - t0, ("reference" if use_ref else "port")


class C4:
    """configs[3]: the WebP lossy post-entropy chain -- residual (dequant + WHT + 4x4 IDCT) -> intra prediction + residual add ->
    loop filter -> YUV420 -> BGRA -- on batches of 1080p key frames made of 16 distinct synthetic frames (uniformly random modes)
    or of copies of a real encoder's frame (tests/golden/webp_file_1080p.npz), tiled to the batch size on the device."""
    c, r, U = 120, 68, 16

    def __init__(self, L, dev, stream, T):
        self.L, self.dev, self.stream, self.T = L, dev, stream, T
        c, r, U = self.c, self.r, self.U
        self.n_mb = n_mb = c * r
        self.q = synth.vp8_quant(seed=2)
        self.filt = synth.vp8_filters(seed=2)
        lv, info, modes = [], [], []
        for i in range(U):
            a, b = synth.vp8_macroblocks(n_mb, seed=100 + i)
            m = synth.vp8_modes(c, r, seed=100 + i)
            m[:, 18] = b[:, 26]
            b[:, 25] = m[:, 0] != 4                 # a Y2 block exactly when the macroblock is not B_PRED
            lv.append(a); info.append(b); modes.append(m)
        self.lv, self.info, self.modes = np.stack(lv), np.stack(info), np.ascontiguousarray(np.stack(modes))
        self.d_q = torch.from_numpy(self.q.astype(np.int16)).to(dev)
        self.d_filt = torch.from_numpy(self.filt).to(dev)
        self.u_lv = torch.from_numpy(self.lv.reshape(U * n_mb, 400)).to(dev)
        self.u_info = torch.from_numpy(self.info.reshape(U * n_mb, 32)).to(dev)
        self.u_modes = torch.from_numpy(self.modes.reshape(U * n_mb, 20)).to(dev)
        self.enc = None
        fx = os.path.join(ROOT, "tests", "golden", "webp_file_1080p.npz")
        if os.path.exists(fx):
            g = np.load(fx)
            e_filt = np.zeros((4, 2, 3), np.uint8)
            ft = C.c_int(-1)
            lfv, lh = g["lf"], g["lf_header"]
            # lf = [level, filter-type bit, segmentation_enabled, triples...], lf_header = [sharpness, segment_feature_mode, lf_update_value[4],
            # adj_enable, mode_ref delta 0, mb_mode delta 0, partitions] as the recorder wrote them (tests/golden/make_golden.py)
            hdr = capi.Vp8FilterHeader(int(lfv[1]), int(lfv[0]), int(lh[0]), int(lfv[2]), int(lh[1]), (C.c_int8 * 4)(*[int(x) for x in lh[2:6]]),
                                       int(lh[6]), int(lh[7]), int(lh[8]), int(lh[9]))
            capi.check(L.ffhip_vp8_filter_params(C.byref(hdr), e_filt.ctypes.data, C.byref(ft)))
            self.enc = {"modes": np.ascontiguousarray(g["modes"]), "u_modes": torch.from_numpy(np.ascontiguousarray(g["modes"])).to(dev),
                        "u_res": torch.from_numpy(np.ascontiguousarray(g["residual"])).to(dev), "d_filt": torch.from_numpy(e_filt).to(dev), "ft": ft.value,
                        "row_sums": g["bgra_row_sums"], "level": int(lfv[0]), "bpred": round(float((g["modes"][:, 0] == 4).mean()) * 100)}

    def batch(self, nf, source):
        """Device buffers and stage closures for `nf` frames of `source` ("random" | "encoder")."""
        L, dev, stream, c, r, n_mb, U = self.L, self.dev, self.stream, self.c, self.r, self.n_mb, self.U
        reps = (nf + U - 1) // U
        B = type("Batch", (), {})()
        B.nf, B.Wp, B.Hp = nf, 16 * c, 16 * r
        d_lv = self.u_lv.repeat(reps, 1)[:nf * n_mb]
        d_info = self.u_info.repeat(reps, 1)[:nf * n_mb]
        d_res = torch.empty((nf * n_mb, 384), dtype=torch.int16, device=dev)
        if source == "encoder":
            h_modes = np.ascontiguousarray(np.broadcast_to(self.enc["modes"], (nf,) + self.enc["modes"].shape))
            d_modes = self.enc["u_modes"].repeat(nf, 1)
            p_res = self.enc["u_res"].repeat(nf, 1)        # the residual the reference's decoder recorded; s_res still runs, on the synthetic levels
            d_filt, ft = self.enc["d_filt"], self.enc["ft"]
        else:
            h_modes = np.ascontiguousarray(np.tile(self.modes, (reps, 1, 1))[:nf])
            d_modes = self.u_modes.repeat(reps, 1)[:nf * n_mb]
            p_res, d_filt, ft = d_res, self.d_filt, 2
        if os.environ.get("BENCH_VP8_FT"):                 # diagnostics (tests/tools/bench_vp8_frames.py): what the loop filter costs the chain
            ft = int(os.environ["BENCH_VP8_FT"])
        Y = torch.zeros((nf, B.Hp, B.Wp), dtype=torch.uint8, device=dev)
        U_ = torch.zeros((nf, B.Hp // 2, B.Wp // 2), dtype=torch.uint8, device=dev)
        V = torch.zeros_like(U_)
        B.bgra = torch.empty((nf, B.Hp, B.Wp * 4), dtype=torch.uint8, device=dev)
        B.planes = (Y, U_, V)
        B.keep = (d_lv, d_info, d_res, d_modes, p_res, h_modes)
        B.s_res = lambda: capi.check(L.ffhip_vp8_residual_batch(nf * n_mb, d_lv.data_ptr(), d_info.data_ptr(), self.d_q.data_ptr(), d_res.data_ptr(), stream))
        B.s_pred = lambda: capi.check(L.ffhip_vp8_predict_recon(c, r, nf, h_modes.ctypes.data, d_modes.data_ptr(), p_res.data_ptr(), n_mb * 384, None, Y.data_ptr(),
                                                                 U_.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, stream))
        B.s_lf = lambda: capi.check(L.ffhip_vp8_loopfilter(c, r, nf, ft, d_modes.data_ptr(), d_filt.data_ptr(), Y.data_ptr(), U_.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, stream))
        # prediction and loop filter as ONE call: the two row kernels side by side
        B.s_pred_lf = lambda: capi.check(L.ffhip_vp8_predict_loopfilter(c, r, nf, h_modes.ctypes.data, d_modes.data_ptr(), p_res.data_ptr(), n_mb * 384, None, ft, d_filt.data_ptr(),
                                                                         Y.data_ptr(), U_.data_ptr(), V.data_ptr(), 256 * n_mb, 64 * n_mb, stream))
        B.s_col = lambda: capi.check(L.ffhip_yuv420_to_bgra(B.bgra.data_ptr(), B.Wp * 4, Y.data_ptr(), U_.data_ptr(), V.data_ptr(), B.Wp, B.Wp // 2, r, c, nf, B.Hp * B.Wp,
                                                            B.Hp * B.Wp // 4, B.Hp * B.Wp * 4, stream))

        # ... and prediction, loop filter and colour conversion as ONE call (round 4): ffhip_vp8_decode_frames -- one fused kernel (a workgroup per
        # frame, every pixel stored once) for chip-filling batches, the row kernels + the colour kernel (on library scratch planes) for small ones
        B.s_frames = lambda: capi.check(L.ffhip_vp8_decode_frames(c, r, nf, h_modes.ctypes.data, d_modes.data_ptr(), p_res.data_ptr(), n_mb * 384, None, ft, d_filt.data_ptr(),
                                                                  B.bgra.data_ptr(), B.Wp * 4, B.Hp * B.Wp * 4, None, None, None, 0, 0, stream))

        def chain_stages():
            B.s_res(); B.s_pred_lf(); B.s_col()
        B.chain_stages = chain_stages

        def chain():
            B.s_res(); B.s_frames()
        B.chain = chain

        def one_clean_pass():
            for p in B.planes:                        # the reference's wrapped 16x16 H_PRED / V_PRED at the frame edge read what the planes
                p.zero_()                             # held before the frame (predict.c:338-353): zeros, as in its freshly allocated planes
            chain()
            capi.check(L.ffhip_stream_sync(stream))
        B.one_clean_pass = one_clean_pass
        return B

    def parity(self, B, source, frames):
        """Frames `frames` of the batch's BGRA against the reference: its whole-file decode (encoder) or its C chain (random)."""
        ok = True
        for i in frames:
            got = B.bgra[i].cpu().numpy()
            if source == "encoder":
                rows = got.reshape(B.Hp, -1).view(np.uint32).astype(np.uint64)
                sums = (rows * (np.arange(rows.shape[1], dtype=np.uint64) + np.uint64(1))).sum(axis=1, dtype=np.uint64)
                ok = ok and bool(np.array_equal(sums, self.enc["row_sums"]))
            else:
                k = i % self.U
                exp, _, _ = c4_cpu_chain_frame(self.c, self.r, self.lv[k], self.info[k], self.q, self.modes[k], 2, self.filt)
                ok = ok and bool(np.array_equal(got, exp))
        return ok

    def sweep(self, sizes, parity_at_largest=True):
        """Chain throughput against the number of frames in the call (the frame loop this replaces is webp.c:1833-1866, one frame
        at a time): a frame's chain is a dependency chain of ~400 macroblock steps, so one frame keeps a handful of the chip's
        1024 SIMDs busy and a batch's frames run side by side."""
        out = {}
        for source in ("encoder", "random") if self.enc else ("random",):
            rows = []
            for nf in sizes:
                B = self.batch(nf, source)
                reps = 5 if nf <= 64 else 3
                ms = self.T.ms(B.chain, reps=reps, warm=1)
                pl = self.T.ms(B.s_pred_lf, reps=reps, warm=0)
                fr = self.T.ms(B.s_frames, reps=reps, warm=0)
                t0 = time.perf_counter()
                B.s_frames()
                host_ms = (time.perf_counter() - t0) * 1e3          # what the enqueue call itself costs the host
                row = {"frames": nf, "chain_ms": round(ms, 4), "value": round(nf * B.Hp * B.Wp / ms / 1e3, 1), "unit": "Mpixels/s",
                       "decode_frames_ms": round(fr, 4), "form": "fused" if self.L.ffhip_vp8_decode_frames_form(nf) == 1 else "rows",
                       "row_kernels_predict+loopfilter_ms": round(pl, 4), "host_enqueue_ms": round(host_ms, 3)}
                if parity_at_largest and nf == max(sizes):
                    B.one_clean_pass()
                    row["parity_first_and_last_frame"] = self.parity(B, source, sorted({0, nf - 1}))
                rows.append(row)
                del B
                torch.cuda.empty_cache()
            out[source] = rows
        return out


def extra_c4(L, dev, stream, T, cpu=True, sweep_sizes=(1, 16, 64, 256, 1024)):
    X = C4(L, dev, stream, T)
    c, r, n_mb, nf = X.c, X.r, X.n_mb, 16
    B = X.batch(nf, "random")
    px = nf * B.Hp * B.Wp
    chain_ms = T.ms(B.chain, reps=5, warm=2)
    stages = {}
    for name, fn, nbytes in (("residual", B.s_res, nf * n_mb * (800 + 32 + 768)), ("predict_recon", B.s_pred, nf * n_mb * (768 + 20 + 384)),
                             ("loopfilter", B.s_lf, nf * n_mb * (2 * 384 + 20)), ("yuv420_to_bgra", B.s_col, px * 5.5)):
        ms = T.ms(fn, reps=5, warm=1)
        stages[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1), "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes": int(nbytes)}
    stages["predict_recon"]["bound"] = stages["loopfilter"]["bound"] = "dependency chain (a wave per macroblock row), not HBM"
    stages["predict_recon+loopfilter_side_by_side"] = {"ms": round(T.ms(B.s_pred_lf, reps=5, warm=1), 4),
                                                       "note": "ffhip_vp8_predict_loopfilter, what the chain calls: the two stages above as one call, their row kernels overlapping"}
    B.one_clean_pass()                                # leave the planes as ONE pass of the chain makes them
    res = {"workload": "C4: 16 x 1920x1088 VP8 key frames, residual -> predict -> loop filter (normal) -> BGRA", "chain_ms": round(chain_ms, 4),
           "value": round(px / chain_ms / 1e3, 1), "unit": "Mpixels/s", "stages": stages,
           # the chain's dominant kernel is the prediction; it is bound by its dependency chain, the HBM figure only shows how far from the roofline that leaves it
           "roofline": dict(roof(nf * n_mb * (768 + 20 + 384), stages["predict_recon"]["ms"]), kernel="k_vp8_predict_rows",
                            note="dominant kernel of the chain at 16 frames; dependency-bound (one wave per macroblock row): see batch_sweep for what the chip does with more frames in flight")}
    if cpu:
        exp, dt, kind = c4_cpu_chain_frame(c, r, X.lv[0], X.info[0], X.q, X.modes[0], 2, X.filt)
        exp2, dt2, _ = c4_cpu_chain_frame(c, r, X.lv[nf - 1], X.info[nf - 1], X.q, X.modes[nf - 1], 2, X.filt)
        res["parity_vs_reference_first_and_last_frame"] = bool(np.array_equal(B.bgra[0].cpu().numpy(), exp) and np.array_equal(B.bgra[nf - 1].cpu().numpy(), exp2))
        res["cpu_baseline"] = {"value": round(2 * B.Hp * B.Wp / (dt + dt2) / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                               "sample": "frames 0 and 15 of the batch (1920x1088 each) through the same four stages in C, one call per frame "
                                         "(ref_vp8_chain_frame: the reference's own functions looped like vp8_decode)"}
        # all host cores: frames over threads (ctypes releases the GIL), two frames per thread
        from concurrent.futures import ThreadPoolExecutor
        cores = max(1, min(os.cpu_count() or 1, 16))
        with ThreadPoolExecutor(cores) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda k: [c4_cpu_chain_frame(c, r, X.lv[(k + j) % nf], X.info[(k + j) % nf], X.q, X.modes[(k + j) % nf], 2, X.filt) for j in range(2)], range(cores)))
            dta = time.perf_counter() - t0
        res["cpu_baseline_all_cores"] = {"value": round(2 * cores * B.Hp * B.Wp / dta / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": kind,
                                         "sample": f"{2 * cores} frames, two per thread"}
    del B
    torch.cuda.empty_cache()
    if X.enc:
        # The same chain on a REAL encoder's syntax elements (tests/golden/webp_file_1080p.npz: libwebp on a photograph mosaic,
        # decoded by the reference; 16 copies of the frame): the uniformly random modes above put H_PRED into column 0 of one
        # row in five, where the reference's wrapped read chains the row to the END of the row above -- libwebp never
        # chooses it there.  Checked against the reference's own whole-file decode (per-row checksums of its BGRA).
        E = X.batch(nf, "encoder")
        e_ms = T.ms(E.chain, reps=5, warm=2)
        e_stage = {"predict_recon_ms": round(T.ms(E.s_pred, reps=5, warm=1), 4), "loopfilter_ms": round(T.ms(E.s_lf, reps=5, warm=1), 4),
                   "predict_recon+loopfilter_side_by_side_ms": round(T.ms(E.s_pred_lf, reps=5, warm=1), 4)}
        E.one_clean_pass()
        res["encoder_stream"] = {"workload": "the same chain on 16 copies of a libwebp-encoded 1920x1088 photograph mosaic (quality 75: loop filter level "
                                             f"{X.enc['level']}, {X.enc['bpred']} % B_PRED macroblocks), syntax elements as the reference's decoder recorded them",
                                 "chain_ms": round(e_ms, 4), "value": round(px / e_ms / 1e3, 1), "unit": "Mpixels/s", **e_stage,
                                 "parity_vs_reference_whole_file_decode": X.parity(E, "encoder", (0, nf - 1))}
        del E
        torch.cuda.empty_cache()
    if sweep_sizes:
        res["batch_sweep"] = {"workload": "the chain (ffhip_vp8_residual_batch -> ffhip_vp8_decode_frames: the fused frame kernel from half the device's compute units in frames on (128 on an MI355X; ffhip_vp8_decode_frames_form), the row kernels + colour "
                                          "kernel below) against the number of 1920x1088 frames in ONE call each; `encoder`: copies of the libwebp frame, `random`: 16 distinct "
                                          "frames of uniformly random modes, tiled",
                              **X.sweep(sweep_sizes)}
        # the fused frame kernel's own roofline at its largest encoder-stream batch: residual + modes in, BGRA out, every byte once
        rows = [r for r in res["batch_sweep"].get("encoder", res["batch_sweep"].get("random", [])) if r["form"] == "fused"]
        if rows:
            r = rows[-1]
            res["fused_roofline"] = dict(roof(r["frames"] * n_mb * (768 + 20 + 1024), r["decode_frames_ms"]), kernel="k_vp8_frames", frames=r["frames"],
                                         note="algorithmic bytes per macroblock: 768 residual + 20 mode bytes in, 1024 BGRA out; bound by instruction issue (VALU ~65 % busy), not HBM")
    return res


def hevc_chain_inputs(W, H, seed, tus=None):
    """One intra picture with SURVEY 8d's config-5 TU mix: the TU list, quantised levels grouped by TU size (as a decoder
    would hand them to ffhip_hevc_residual_batch) and residual offsets into one buffer laid out [32x32 | 16x16 | 8x8 | 4x4]."""
    if tus is None:
        tus, _ = synth.hevc_intra_tus(W, H, seed=seed, tu_mix="c5")
    tus = tus.copy()
    tus["flags"] &= ~np.uint8(synth.TU_RDPCM)      # rdpcm belongs to transform-skip / bypass TUs: keep the chain plain
    rng = np.random.default_rng(seed)
    has = (tus["flags"] & synth.TU_RESIDUAL) != 0
    groups, off = {}, 0
    for n in (32, 16, 8, 4):
        idx = np.nonzero(has & (tus["log2_size"] == int(np.log2(n))))[0]
        if idx.size == 0:
            continue
        tus["res_offset"][idx] = off + np.arange(idx.size, dtype=np.uint32) * (n * n)
        lv = np.rint(rng.laplace(0, 6, size=(idx.size, n * n))).astype(np.int16)
        info = np.zeros((idx.size, 4), np.uint8)
        info[:, 0] = 27                                                      # qP of SURVEY 8d C5
        info[:, 1] = ((tus["cidx"][idx] == 0) & (n == 4)).astype(np.uint8)   # intra luma 4x4 takes the DST
        groups[n] = (idx, lv, info, off)
        off += idx.size * n * n
    return tus, groups, off


def run_hevc_chain_gpu(L, dev, stream, W, H, tus, groups, total, T=None, tile_first=None, fused_colour=False):
    d_tus = torch.from_numpy(tus.view(np.uint8).copy()).to(dev)
    d_res = torch.zeros(total + 64, dtype=torch.int16, device=dev)
    dg = {n: (torch.from_numpy(lv).to(dev), torch.from_numpy(info).to(dev), off, len(idx)) for n, (idx, lv, info, off) in groups.items()}
    py = torch.zeros((H, W), dtype=torch.int16, device=dev)
    pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev)
    pv = torch.zeros_like(pu)
    bgra = torch.empty((H, W * 4), dtype=torch.uint8, device=dev)

    def s_res():
        for n, (lv, info, off, cnt) in dg.items():
            capi.check(L.ffhip_hevc_residual_batch(n, cnt, lv.data_ptr(), info.data_ptr(), None, 8, 0, d_res.data_ptr() + 2 * off, stream))

    tf = None if tile_first is None else np.ascontiguousarray(tile_first, dtype=np.int64)

    def s_intra():
        if tf is not None:   # the tile loop as one pipelined call: the list is the concatenation of independent tiles
            capi.check(L.ffhip_hevc_intra_recon_tiles(tus.ctypes.data, d_tus.data_ptr(), len(tus), tf.ctypes.data, len(tf), d_res.data_ptr(), py.data_ptr(), pu.data_ptr(),
                                                      pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, stream))
            return
        capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, d_tus.data_ptr(), len(tus), d_res.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(),
                                            W, H, W, W // 2, H // 2, W // 2, 8, 8, stream))

    def s_col():
        capi.check(L.ffhip_yuv420_to_bgra_16(bgra.data_ptr(), W * 4, py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, W // 2, H // 64, W // 64, 64, 1,
                                             H * W, H * W // 4, H * W * 4, stream))

    def s_decode():      # intra reconstruction + colour conversion as ONE call (ffhip_hevc_decode_tiles)
        capi.check(L.ffhip_hevc_decode_tiles(tus.ctypes.data, d_tus.data_ptr(), len(tus), tf.ctypes.data, len(tf), d_res.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(),
                                             W, H, W, W // 2, H // 2, W // 2, 8, 8, bgra.data_ptr(), W * 4, stream))

    def chain():
        if fused_colour:
            s_res(); s_decode()
        else:
            s_res(); s_intra(); s_col()
    times = None
    if T is not None:
        times = {"chain": T.ms(chain, reps=5, warm=2), "residual": T.ms(s_res, reps=5, warm=1), "intra_recon": T.ms(s_intra, reps=5, warm=1),
                 "yuv420_to_bgra_16": T.ms(s_col, reps=5, warm=1)}
        t0 = time.perf_counter()
        s_intra()
        times["intra_host_enqueue"] = (time.perf_counter() - t0) * 1e3
    for p in (py, pu, pv):
        p.zero_()
    chain()
    capi.check(L.ffhip_stream_sync(stream))
    if times is not None:      # what the device planner made of the list: taken, wavefront tickets, window, sorted by plane
        v = (C.c_uint32 * 8)()
        if L.ffhip_debug_hevc_plan_result(v) == 0:
            times["plan"] = {"taken": v[0] == 0, "wavefront_tickets": v[3] == 0, "groups": int(v[1]), "width": int(v[4]), "window_log2": int(v[5]), "sorted_by_plane": bool(v[7])}
    return bgra, (py, pu, pv), d_res, times


def c5_cpu_chain_picture(W, H, tus, groups, total):
    """ONE picture through scaling + inverse transforms -> intra prediction + reconstruction -> BGRA in C, one call
    (oracle/_ref's ref_hevc_chain_picture: the reference's own functions; oracle/ffo_chain.c when it did not travel)."""
    O = oracle_lib()
    use_ref = os.path.exists(O.REF_SO)
    lib = O.ref() if use_ref else O.ffo()
    fn = lib.ref_hevc_chain_picture if use_ref else lib.ffo_hevc_chain_picture
    vp = C.c_void_p
    fn.argtypes = [vp, C.c_long, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    fn.restype = None
    levels = np.zeros(total + 64, np.int16)
    for n, (idx, lv, info, off) in groups.items():
        levels[off:off + lv.size] = lv.reshape(-1)
    resid = np.zeros(total + 64, np.int16)
    py, pu, pv = np.zeros((H, W), np.int16), np.zeros((H // 2, W // 2), np.int16), np.zeros((H // 2, W // 2), np.int16)
    o = np.zeros((H, W * 4), np.uint8)
    t = np.ascontiguousarray(tus)
    t0 = time.perf_counter()
    fn(t.ctypes.data, len(t), levels.ctypes.data, resid.ctypes.data, 27, 8, py.ctypes.data, pu.ctypes.data, pv.ctypes.data, W, H, 64, o.ctypes.data, W * 4)
    return o, resid, time.perf_counter() - t0, ("reference" if use_ref else "port")


def c5_grid_sweep(L, dev, stream, T, cpu=True, pictures=(1, 4, 8), tile=512, tiles_xy=(15, 9)):
    """configs[4] says "single 8K tile grid": an 8K picture as a HEIF grid of 15 x 9 = 135 independent 512x512 tiles (7680x4608), every
    tile a complete intra picture of the config-5 TU mix with its own neighbour availability, all tiles of all pictures in ONE plane set and
    ONE ffhip_hevc_residual_batch / ffhip_hevc_intra_recon / ffhip_yuv420_to_bgra_16 call each (the tile loop this replaces decodes them
    one after the other: heif.c:297-309).  Pictures side by side: 1, 4 (2 x 2), 8 (4 x 2)."""
    t0, _ = synth.hevc_intra_tus(tile, tile, seed=3, tu_mix="c5")
    _, perm0 = synth.hevc_reference_order(t0, 64, 2, 3, return_perm=True)
    out = {"workload": f"8K pictures as grids of {tiles_xy[0]} x {tiles_xy[1]} independent {tile}x{tile} HEVC tiles ({len(t0)} TUs per tile, config-5 mix), "
                       "residual batches -> ffhip_hevc_decode_tiles (the tile loop with its colour conversion as one call: the pre-pass off the stream, on two scratch sets in turn); "
                       "`pipelined_three_calls`: ffhip_hevc_intra_recon_tiles + ffhip_yuv420_to_bgra_16; `one_call_unpipelined`: ffhip_hevc_intra_recon over the whole list + the colour kernel", "rows": []}
    for npic in pictures:
        px_, py_ = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}[npic]
        gx, gy = tiles_xy[0] * px_, tiles_xy[1] * py_
        W, H = gx * tile, gy * tile
        tus = np.tile(t0, gx * gy)
        k = np.repeat(np.arange(gx * gy), len(t0))
        sc = np.where(tus["cidx"] == 0, tile, tile // 2)
        tus["x"] = (tus["x"].astype(np.int64) + (k % gx) * sc).astype(np.uint16)
        tus["y"] = (tus["y"].astype(np.int64) + (k // gx) * sc).astype(np.uint16)
        tus, groups, total = hevc_chain_inputs(W, H, seed=40 + npic, tus=tus)
        # the same tiles, every tile's list in the order the reference decodes it in (per coding unit: luma tree, Cb, Cr; coding/hevc.c:5013-5180)
        tus_r = tus[(np.arange(gx * gy, dtype=np.int64)[:, None] * len(t0) + perm0[None, :]).reshape(-1)]
        tile_first = np.arange(gx * gy, dtype=np.int64) * len(t0)
        bgra_1, _, _, t1 = run_hevc_chain_gpu(L, dev, stream, W, H, tus, groups, total, T)                 # one ffhip_hevc_intra_recon call over the whole list
        bgra_r, _, _, tr = run_hevc_chain_gpu(L, dev, stream, W, H, tus_r, groups, total, T, tile_first, fused_colour=True)
        bgra_3, _, _, t3 = run_hevc_chain_gpu(L, dev, stream, W, H, tus, groups, total, T, tile_first)    # ffhip_hevc_intra_recon_tiles + ffhip_yuv420_to_bgra_16
        bgra, planes, d_res, t = run_hevc_chain_gpu(L, dev, stream, W, H, tus, groups, total, T, tile_first, fused_colour=True)   # ffhip_hevc_decode_tiles: the pipelined tile loop with its colour conversion
        row = {"pictures": npic, "tiles": gx * gy, "tus": int(len(tus)), "chain_ms": round(t["chain"], 4), "intra_recon_ms": round(t["intra_recon"], 4),
               "intra_host_enqueue_ms": round(t["intra_host_enqueue"], 3), "value": round(W * H / t["chain"] / 1e3, 1), "unit": "Mpixels/s", "plan": t.get("plan"),
               "reference_order": {"chain_ms": round(tr["chain"], 4), "intra_recon_ms": round(tr["intra_recon"], 4), "intra_host_enqueue_ms": round(tr["intra_host_enqueue"], 3),
                                   "value": round(W * H / tr["chain"] / 1e3, 1), "plan": tr.get("plan"), "same_pixels": bool(torch.equal(bgra, bgra_r))},
               "one_call_unpipelined": {"chain_ms": round(t1["chain"], 4), "intra_recon_ms": round(t1["intra_recon"], 4), "value": round(W * H / t1["chain"] / 1e3, 1),
                                        "same_pixels": bool(torch.equal(bgra, bgra_1))},
               "pipelined_three_calls": {"chain_ms": round(t3["chain"], 4), "value": round(W * H / t3["chain"] / 1e3, 1), "same_pixels": bool(torch.equal(bgra, bgra_3))}}
        del bgra_r, bgra_1, bgra_3
        if cpu and npic == max(pictures):
            # parity of the first and the last tile of the largest grid: each tile is a picture of its own for the reference's C chain
            ok = True
            for ti in (0, gx * gy - 1):
                sl = slice(ti * len(t0), (ti + 1) * len(t0))
                tt = tus[sl].copy()
                ox, oy = (ti % gx) * tile, (ti // gx) * tile
                tt["x"] = (tt["x"].astype(np.int64) - np.where(tt["cidx"] == 0, ox, ox // 2)).astype(np.uint16)
                tt["y"] = (tt["y"].astype(np.int64) - np.where(tt["cidx"] == 0, oy, oy // 2)).astype(np.uint16)
                # the tile's levels, re-packed as a picture of its own
                lv_all = {n: (g[1], g[3], {int(i): j for j, i in enumerate(g[0])}) for n, g in groups.items()}
                tg, toff = {}, 0
                for n in (32, 16, 8, 4):
                    if n not in lv_all:
                        continue
                    lv, off, pos = lv_all[n]
                    idx = np.nonzero(((tt["flags"] & synth.TU_RESIDUAL) != 0) & (tt["log2_size"] == int(np.log2(n))))[0]
                    if idx.size == 0:
                        continue
                    rows_ = np.array([pos[int(sl.start + i)] for i in idx])
                    tt["res_offset"][idx] = toff + np.arange(idx.size, dtype=np.uint32) * (n * n)
                    tg[n] = (idx, lv[rows_], None, toff)
                    toff += idx.size * n * n
                o, _, _, _ = c5_cpu_chain_picture(tile, tile, tt, tg, toff)
                got = bgra[oy:oy + tile].cpu().numpy().reshape(tile, W, 4)[:, ox:ox + tile].reshape(tile, tile * 4)
                ok = ok and bool(np.array_equal(got, o))
            row["parity_first_and_last_tile_vs_reference"] = ok
        out["rows"].append(row)
        del bgra, planes, d_res
        torch.cuda.empty_cache()
    return out


def extra_c5(L, dev, stream, T, cpu=True, grid=True):
    """configs[4]: HEIF/HEVC still, one 8K picture (7680x4352 coded): scaling + inverse transforms per TU size -> intra
    prediction + reconstruction -> YUV420 16-bit -> BGRA"""
    W, H = 7680, 4352
    tus, groups, total = hevc_chain_inputs(W, H, seed=5)
    _, _, _, t = run_hevc_chain_gpu(L, dev, stream, W, H, tus, groups, total, T)
    px = W * H
    n_res = sum(len(g[0]) * n * n for n, g in groups.items())
    stages = {"residual": {"ms": round(t["residual"], 4), "GB/s": round(4 * n_res / t["residual"] / 1e6, 1), "algorithmic_bytes": 4 * n_res,
                           "frac_of_hbm_peak": round(4 * n_res / t["residual"] / 1e6 / HBM_PEAK_GBS, 4),
                           "launches": {f"{n}x{n}": int(len(g[0])) for n, g in groups.items()}},
              "intra_recon": {"ms": round(t["intra_recon"], 4), "GB/s": round(9 * px / t["intra_recon"] / 1e6, 1), "algorithmic_bytes": int(9 * px),
                              "frac_of_hbm_peak": round(9 * px / t["intra_recon"] / 1e6 / HBM_PEAK_GBS, 5), "tus": int(len(tus)),
                              "bound": "dependency chain (a wave per 64x64 window group), not HBM"},
              "yuv420_to_bgra_16": {"ms": round(t["yuv420_to_bgra_16"], 4), "GB/s": round(7 * px / t["yuv420_to_bgra_16"] / 1e6, 1),
                                    "algorithmic_bytes": int(7 * px), "frac_of_hbm_peak": round(7 * px / t["yuv420_to_bgra_16"] / 1e6 / HBM_PEAK_GBS, 4)}}
    # the same picture from the list in the reference's order (per coding unit the luma tree, then Cb, then Cr: coding/hevc.c:5013-5180)
    bg_p, _, _, _ = run_hevc_chain_gpu(L, dev, stream, W, H, tus, groups, total)
    bg_r, _, _, tr = run_hevc_chain_gpu(L, dev, stream, W, H, synth.hevc_reference_order(tus, 64, 2, 5), groups, total, T)
    same = bool(torch.equal(bg_p, bg_r))
    # ... and as a one-tile call of ffhip_hevc_intra_recon_tiles: the pre-pass does not wait for the stream (the TU list is complete when the call is
    # made), so in a decoder's loop over pictures it runs next to the colour conversion of the picture before and this picture's residual batches
    bg_t, _, _, tt = run_hevc_chain_gpu(L, dev, stream, W, H, tus, groups, total, T, tile_first=np.zeros(1, np.int64), fused_colour=True)
    same_t = bool(torch.equal(bg_p, bg_t))
    del bg_p, bg_r, bg_t
    res = {"workload": "C5: one 7680x4352 HEVC intra picture, TU mix of SURVEY 8d (luma 32/16 at 60/40, chroma 16/8), qP 27", "chain_ms": round(t["chain"], 4),
           "value": round(px / t["chain"] / 1e3, 1), "unit": "Mpixels/s", "stages": stages, "plan": t.get("plan"),
           "reference_order": {"chain_ms": round(tr["chain"], 4), "value": round(px / tr["chain"] / 1e3, 1), "intra_recon_ms": round(tr["intra_recon"], 4),
                               "intra_host_enqueue_ms": round(tr["intra_host_enqueue"], 3), "plan": tr.get("plan"), "same_pixels": same,
                               "note": "the same TUs, records interleaved per coding unit (luma tree, Cb, Cr) as decode_cu_coded_intra_prediction_mode walks them; "
                                       "`value` above: each coding tree block's planes one after the other"},
           "pipelined": {"chain_ms": round(tt["chain"], 4), "value": round(px / tt["chain"] / 1e3, 1), "intra_recon_ms": round(tt["intra_recon"], 4), "same_pixels": same_t,
                         "note": "ffhip_hevc_decode_tiles with one tile: the pre-pass on the library's stream, not waiting for `stream`; pictures back to back"},
           "roofline": dict(roof(9 * px, t["intra_recon"]), kernel="k_hevc_intra_groups",
                            note="algorithmic bytes 9 B/pixel (3 + 3 in, 3 out); the stage is bound by its dependency chain")}
    if cpu:
        # bounded sample: a 1024x512 picture of the same mix through the reference's C (one call), and the same picture through the GPU chain
        sw, sh = 1024, 512
        stus, sgroups, stotal = hevc_chain_inputs(sw, sh, seed=6)
        g_bgra, g_planes, g_res, _ = run_hevc_chain_gpu(L, dev, stream, sw, sh, stus, sgroups, stotal)
        o, resid, dt, kind = c5_cpu_chain_picture(sw, sh, stus, sgroups, stotal)
        res["parity_vs_reference_sample"] = bool(np.array_equal(g_bgra.cpu().numpy(), o) and np.array_equal(g_res[:stotal].cpu().numpy(), resid[:stotal]))
        res["cpu_baseline"] = {"value": round(sw * sh / dt / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": kind,
                               "sample": f"one {sw}x{sh} picture of the same TU mix ({len(stus)} TUs) through the same three stages in C, one call per picture "
                                         "(ref_hevc_chain_picture: the reference's own functions per TU in decode order)"}
        from concurrent.futures import ThreadPoolExecutor
        cores = max(1, min(os.cpu_count() or 1, 16))
        with ThreadPoolExecutor(cores) as ex:      # all host cores: the same picture once per thread (pictures are independent)
            t0 = time.perf_counter()
            list(ex.map(lambda k: c5_cpu_chain_picture(sw, sh, stus, sgroups, stotal), range(cores)))
            dta = time.perf_counter() - t0
        res["cpu_baseline_all_cores"] = {"value": round(cores * sw * sh / dta / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": kind,
                                         "sample": f"{cores} pictures of {sw}x{sh}, one per thread"}
    if grid:
        res["grid"] = c5_grid_sweep(L, dev, stream, T, cpu=cpu)
    return res

_REF_LOAD_CHILD = r"""
import ctypes as C, hashlib, sys, time
path, so, out = sys.argv[1], sys.argv[2], sys.argv[3]
R = C.CDLL(so, mode=C.RTLD_GLOBAL)
class Pic(C.Structure):      # struct pic, format/file.h:29-40 (leading fields)
    _fields_ = [("pixels", C.c_void_p), ("left", C.c_int), ("top", C.c_int), ("width", C.c_int), ("height", C.c_int), ("depth", C.c_int), ("pitch", C.c_int)]
R.file_ops_init.restype = None
R.file_probe.restype = C.c_void_p
R.file_probe.argtypes = [C.c_char_p]
R.file_load.restype = C.POINTER(Pic)
R.file_load.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
R.file_ops_init()
ops = R.file_probe(path.encode())
t0 = time.perf_counter()
p = R.file_load(ops, path.encode(), 0).contents
dt = time.perf_counter() - t0
h, hm = hashlib.sha256(), hashlib.sha256()      # all rows; and all but the last 16x16 MCU (the reference's bit reader can run dry in a scan's very last data unit)
buf = (C.c_uint8 * (p.height * p.pitch)).from_address(p.pixels)
mv = memoryview(buf)
for y in range(p.height):
    row = mv[y * p.pitch:y * p.pitch + p.width * 4]
    h.update(row)
    hm.update(row[:(p.width - 16) * 4] if y >= p.height - 16 else row)
open(out, "w").write("%d %d %.6f %s %s" % (p.width, p.height, dt, h.hexdigest(), hm.hexdigest()))
sys.stdout.flush()
import os
os._exit(0)     # the reference's loader leaves the heap in a state interpreter teardown does not survive
"""


def f1_reference_load(path, so, copies=1):
    """The reference's own whole-file decode (file_load -> JPG_load: format/jpg.c, coding/huffman.c, utils/idct.c, utils/colorspace.c) of one file,
    in `copies` child processes at once (one load each: its loader is single-threaded and not safe to call twice in a process).  Returns
    (width, height, [seconds per load], (sha256 of the BGRA rows, the same without the last 16x16 MCU))."""
    import subprocess, tempfile
    outs = [tempfile.mktemp(suffix=".txt") for _ in range(copies)]
    procs = [subprocess.Popen([sys.executable, "-c", _REF_LOAD_CHILD, path, so, o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for o in outs]
    for pr in procs:
        pr.wait(timeout=300)
    res = []
    for o in outs:
        if not os.path.exists(o):
            raise RuntimeError("the reference's loader did not finish on " + path)
        w, h, dt, sha, sha_m = open(o).read().split()
        os.unlink(o)
        res.append((int(w), int(h), float(dt), (sha, sha_m)))
    return res[0][0], res[0][1], [r[2] for r in res], res[0][3]


def extra_f1(L, dev, stream, T, cpu=True, n=256):
    """SURVEY 8f row f1, the step in front of the hot path: baseline JPEG FILES in, BGRA in device memory out (ffhip_jpeg_decode_files_device: the
    marker loop, read_dqt, read_compressed_scan / decode_data_unit of format/jpg.c:78-105, 255-415, 588-637 and coding/huffman.c:92-222, then the
    fused reconstruction) on `n` 4K files -- with restart markers (one interval per MCU row: the Huffman decode runs on the device, a lane per
    interval) and without (on the device as well: a lane per 2048-bit subsequence of the scan, synchronised over rounds) --, the device Huffman stage
    alone (ffhip_jpeg_entropy_batch_gpu) with its phases, and the reference's own
    JPG_load on the same file as the CPU baseline.  Wall-clock figures: host code and PCIe are part of this row."""
    import hashlib, io, tempfile
    from PIL import Image
    rng = np.random.default_rng(0)
    W, H = 3840, 2160
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0), 128 + 90 * np.cos(xx / 11.0 + yy / 53.0), (xx * 255 / (W - 1) + yy * 255 / (H - 1)) / 2], axis=2)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    del yy, xx
    vp = C.c_void_p
    threads = max(1, min(os.cpu_count() or 1, 16))
    res = {"workload": f"{n} x {W}x{H} baseline 4:2:0 JPEG files (PIL, quality 85; the same file {n} times) -> BGRA in device memory; wall clock of the calls, "
                       f"{threads} host threads for header parsing / staging", "files": {}}
    geom = capi.JpegGeom()
    bufs = {}
    for tag, kw in (("dri_per_mcu_row", dict(restart_marker_rows=1)), ("no_dri", dict())):
        if os.environ.get("F1_TAGS") and tag not in os.environ["F1_TAGS"].split(","):      # (a row alone, for profiling)
            continue
        bio = io.BytesIO()
        Image.fromarray(img).save(bio, "JPEG", quality=85, subsampling=2, **kw)
        data = bio.getvalue()
        buf = np.frombuffer(data, dtype=np.uint8)
        ptrs = (vp * n)(*([buf.ctypes.data] * n))
        lens = (C.c_size_t * n)(*([buf.size] * n))
        status = (C.c_int * n)()
        d_out = torch.empty((n, H, W * 4), dtype=torch.uint8, device=dev)

        def files_to_pixels():
            capi.check(L.ffhip_jpeg_decode_files_device(ptrs, lens, n, threads, C.byref(geom), d_out.data_ptr(), W * 4, W * 4 * H, status, stream), "ffhip_jpeg_decode_files_device")
            capi.check(L.ffhip_stream_sync(stream))
        files_to_pixels()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); files_to_pixels(); best = min(best, time.perf_counter() - t0)
        row = {"file_bytes": len(data), "files_to_device_pixels_ms": round(best * 1e3, 2), "value": round(n * W * H / best / 1e6, 1), "unit": "Mpixels/s",
               "files_per_s": round(n / best), "compressed_GB/s": round(n * len(data) / best / 1e9, 2)}
        px0 = d_out[0].cpu().numpy()
        sha_gpu = hashlib.sha256(px0.tobytes()).hexdigest()
        sha_gpu_m = hashlib.sha256(px0[:H - 16].tobytes() + px0[H - 16:, :(W - 16) * 4].tobytes()).hexdigest()
        del px0
        if True:
            # the device Huffman stage alone, with its phases (ffhip_debug_huff_times), and the reconstruction of its planes alone
            g = geom
            yb, cb = g.mcu_cols * g.mcu_rows * 4 * 64, g.mcu_cols * g.mcu_rows * 64
            d_y = torch.empty(n * yb, dtype=torch.int16, device=dev)
            d_u = torch.empty(n * cb, dtype=torch.int16, device=dev)
            d_v = torch.empty(n * cb, dtype=torch.int16, device=dev)
            d_q = torch.empty(n * 256, dtype=torch.int16, device=dev)

            def entropy():
                capi.check(L.ffhip_jpeg_entropy_batch_gpu(ptrs, lens, n, threads, C.byref(g), d_y.data_ptr(), d_u.data_ptr(), d_v.data_ptr(), d_q.data_ptr(), status, stream),
                           "ffhip_jpeg_entropy_batch_gpu")
            entropy()
            eb, ph = 1e9, None
            for _ in range(3):
                t0 = time.perf_counter(); entropy(); dt = time.perf_counter() - t0
                if dt < eb:
                    eb = dt
                    tt = (C.c_double * 8)()
                    L.ffhip_debug_huff_times(tt)
                    ph = [float(x) for x in tt]
            rms = T.ms(lambda: capi.check(L.ffhip_jpeg_recon_batch(C.byref(g), n, d_y.data_ptr(), d_u.data_ptr(), d_v.data_ptr(), d_q.data_ptr(), 256, d_out.data_ptr(), W * 4, W * 4 * H,
                                                                   None, 0, stream)), reps=5, warm=1)
            row["entropy_batch_gpu"] = {"ms": round(eb * 1e3, 2), "value": round(n * W * H / eb / 1e6, 1), "unit": "Mpixels/s",
                                        "phases_ms": {"header_parse": round(ph[0] / 1e3, 2), "layout": round(ph[1] / 1e3, 2), "unstuff_and_markers_with_uploads_enqueued": round(ph[2] / 1e3, 2),
                                                      "tables": round(ph[3] / 1e3, 2), "enqueue": round(ph[4] / 1e3, 2), "wait_uploads_clears_kernel": round(ph[5] / 1e3, 2)},
                                        "k_jpeg_huff_ms": round(ph[6] / 1e3, 3), "k_jpeg_huff_compressed_GB/s": round(n * len(data) / (ph[6] / 1e6) / 1e9, 1) if ph[6] else None}
            # no single kernel behind this figure: the batch goes in parts -- bytes up on a copy stream, two rounds over all subsequences, list rounds, scan,
            # write pass -- and the events bracket all of it, the uploads it waits for included (FFHIP_JPEG_SYNC=0: the lane-per-interval kernel alone)
            e = row["entropy_batch_gpu"]
            e["device_pipeline_ms"] = e.pop("k_jpeg_huff_ms")
            e.pop("k_jpeg_huff_compressed_GB/s")
            e["form"] = ("one lane per restart interval (k_jpeg_huff)" if os.environ.get("FFHIP_JPEG_SYNC", "1")[:1] == "0" else
                         "subsequences of 2048 bits of a restart interval (of the scan) a lane, synchronised over rounds (ffhip_huff_gpu.hip, k_huff_span); "
                         "profiles/r5_huff_plain_timeline.txt has the kernels of one call")
            row["reconstruction_ms"] = round(rms, 3)
            del d_y, d_u, d_v, d_q
        if cpu:
            O = oracle_lib()
            if os.path.exists(O.REF_SO):
                try:
                    with tempfile.NamedTemporaryFile(suffix=".jpg", delete=False) as fh:
                        fh.write(data)
                    w1, h1, dts, sha_ref = f1_reference_load(fh.name, O.REF_SO, 1)
                    row["parity_vs_reference_whole_file_decode"] = bool(sha_ref[1] == sha_gpu_m and (w1, h1) == (W, H))
                    if sha_ref[0] != sha_gpu:     # a reference defect kept out of the comparison: its bit reader runs dry in the scan's last data unit
                        row["parity_note"] = "every pixel but the picture's last 16x16 MCU, which the reference's own loader gets wrong on this file (utils/bitstream.c:117)"
                    row["cpu_baseline"] = {"value": round(W * H / min(dts) / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
                                           "sample": "the reference's own whole-file decode (file_load -> JPG_load) of the same file, one load in a process of its own"}
                    cores = threads
                    _, _, dta, _ = f1_reference_load(fh.name, O.REF_SO, cores)
                    row["cpu_baseline_all_cores"] = {"value": round(cores * W * H / max(dta) / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "reference",
                                                     "sample": f"{cores} processes at once, one load each; pixels / the slowest load"}
                    os.unlink(fh.name)
                except Exception as e:
                    row["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        res["files"][tag] = row
        del d_out
        torch.cuda.empty_cache()
        bufs[tag] = buf
    # four times the files, behind BOTH rows of 256 (so that those two are measured in the same state of the process: the library's staging buffers grow for
    # these calls and stay grown).  With restart markers and without alike: the parts are larger, the uploads (PCIe) and the kernels overlap over a longer
    # stretch.  Rows late in the process come out slower than the same row run alone (16 -> 20 ms, 50 -> 70 ms; F1_TAGS=<row> runs one alone): DESIGN.md 5
    for tag, buf in bufs.items():
        if True:
            n4 = 4 * n
            ptrs4 = (vp * n4)(*([buf.ctypes.data] * n4))
            lens4 = (C.c_size_t * n4)(*([buf.size] * n4))
            status4 = (C.c_int * n4)()
            d_out4 = torch.empty((n4, H, W * 4), dtype=torch.uint8, device=dev)

            def files4():
                capi.check(L.ffhip_jpeg_decode_files_device(ptrs4, lens4, n4, threads, C.byref(geom), d_out4.data_ptr(), W * 4, W * 4 * H, status4, stream), "ffhip_jpeg_decode_files_device")
                capi.check(L.ffhip_stream_sync(stream))
            files4()
            b4 = 1e9
            for _ in range(2):
                t0 = time.perf_counter(); files4(); b4 = min(b4, time.perf_counter() - t0)
            tt = (C.c_double * 8)()
            L.ffhip_debug_huff_times(tt)
            res["files"][f"{tag}_x{n4}"] = {"files": n4, "files_to_device_pixels_ms": round(b4 * 1e3, 2), "value": round(n4 * W * H / b4 / 1e6, 1), "unit": "Mpixels/s",
                                            "files_per_s": round(n4 / b4), "device_pipeline_ms": round(float(tt[6]) / 1e3, 3),
                                            "same_pixels_as_first_file": bool(torch.equal(d_out4[0], d_out4[n4 - 1]))}
            del d_out4
            torch.cuda.empty_cache()
    # two callers, each with a stream, output buffer and eight host threads of its own, six calls of 256 files each back to back: one caller's header parsing
    # and staging run under the other's uploads and kernels -- what a service that keeps the device busy sees
    try:
        import threading
        nt2, calls2 = 2, 6
        go, errs, fin = threading.Barrier(nt2 + 1), [], [0.0] * nt2
        buf2 = bufs["no_dri"] if "no_dri" in bufs else next(iter(bufs.values()))

        def caller(k):
            try:
                p2 = (vp * n)(*([buf2.ctypes.data] * n)); l2 = (C.c_size_t * n)(*([buf2.size] * n)); s2 = (C.c_int * n)()
                g2 = capi.JpegGeom()
                st2 = torch.cuda.Stream(device=dev)
                o2 = torch.empty((n, H, W * 4), dtype=torch.uint8, device=dev)

                def one():
                    capi.check(L.ffhip_jpeg_decode_files_device(p2, l2, n, max(1, threads // 2), C.byref(g2), o2.data_ptr(), W * 4, W * 4 * H, s2, st2.cuda_stream), "ffhip_jpeg_decode_files_device")
                    capi.check(L.ffhip_stream_sync(st2.cuda_stream))
                one()
                go.wait()
                for _ in range(calls2):
                    one()
                fin[k] = time.perf_counter()
            except Exception as e:      # noqa: BLE001
                errs.append(f"{type(e).__name__}: {e}")
                go.abort()
        ths = [threading.Thread(target=caller, args=(k,)) for k in range(nt2)]
        for t in ths:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        if errs:
            raise RuntimeError("; ".join(errs))
        dt2 = max(fin) - t0
        res["files"]["two_callers"] = {"callers": nt2, "calls_each": calls2, "files_per_call": n, "ms_per_call_aggregate": round(dt2 * 1e3 / (nt2 * calls2), 2),
                                       "value": round(nt2 * calls2 * n * W * H / dt2 / 1e6, 1), "unit": "Mpixels/s", "files_per_s": round(nt2 * calls2 * n / dt2)}
    except Exception as e:      # noqa: BLE001 -- a row of its own: the others stand
        res["files"]["two_callers"] = {"error": f"{type(e).__name__}: {e}"}
    # many small files: 4 096 thumbnails of 256x256 (64 different ones in turn), where the host's share -- header parsing, table look-up, staging -- is what counts
    try:
        tw = th = 256
        nt = 4096
        thumbs = []
        for i in range(64):
            ty, tx = np.mgrid[0:th, 0:tw]
            timg = np.stack([128 + 100 * np.sin(tx / (9.0 + i)), 128 + 90 * np.cos(ty / (7.0 + i % 5)), (tx * 3 + ty * 5 + i * 7) % 256], axis=2)
            timg = np.clip(timg + rng.normal(0, 20, timg.shape), 0, 255).astype(np.uint8)
            bio = io.BytesIO()
            Image.fromarray(timg).save(bio, "JPEG", quality=80, subsampling=2)
            thumbs.append(np.frombuffer(bio.getvalue(), dtype=np.uint8))
        tptrs = (vp * nt)(*[thumbs[i % 64].ctypes.data for i in range(nt)])
        tlens = (C.c_size_t * nt)(*[thumbs[i % 64].size for i in range(nt)])
        tstatus = (C.c_int * nt)()
        d_t = torch.empty((nt, th, tw * 4), dtype=torch.uint8, device=dev)

        def thumbs_to_pixels():
            capi.check(L.ffhip_jpeg_decode_files_device(tptrs, tlens, nt, threads, C.byref(geom), d_t.data_ptr(), tw * 4, tw * 4 * th, tstatus, stream), "ffhip_jpeg_decode_files_device")
            capi.check(L.ffhip_stream_sync(stream))
        thumbs_to_pixels()
        bt = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); thumbs_to_pixels(); bt = min(bt, time.perf_counter() - t0)
        trow = {"files": nt, "size": f"{tw}x{th}", "file_bytes": int(thumbs[0].size), "files_to_device_pixels_ms": round(bt * 1e3, 2), "value": round(nt * tw * th / bt / 1e6, 1),
                "unit": "Mpixels/s", "files_per_s": round(nt / bt), "same_pixels_64_files_apart": bool(torch.equal(d_t[0], d_t[64]))}
        if cpu:
            O = oracle_lib()
            if os.path.exists(O.REF_SO):
                with tempfile.NamedTemporaryFile(suffix=".jpg", delete=False) as fh:
                    fh.write(thumbs[0].tobytes())
                w1, h1, dts, sha_ref = f1_reference_load(fh.name, O.REF_SO, 1)
                os.unlink(fh.name)
                px0 = d_t[0].cpu().numpy()
                sha_m = hashlib.sha256(px0[:th - 16].tobytes() + px0[th - 16:, :(tw - 16) * 4].tobytes()).hexdigest()
                trow["parity_vs_reference_whole_file_decode"] = bool(sha_ref[1] == sha_m and (w1, h1) == (tw, th))
        res["files"]["thumbnails_x4096"] = trow
        del d_t
        torch.cuda.empty_cache()
    except Exception as e:      # noqa: BLE001 -- a row of its own: the others stand
        res["files"]["thumbnails_x4096"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def compact_configs(extra):
    """Every BASELINE config and layout in a few hundred bytes each: what the result line carries (the verbose `extra` goes to stderr /
    --extra-file).  Values in Mpixels/s, `frac` = fraction of the 8 TB/s HBM peak of the kernel named in the verbose record."""
    def g(d, *ks):
        for k in ks:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    out = {}
    c2 = extra.get("c2", {})
    out["c2"] = {"value": g(c2, "value"), "frac": g(c2, "roofline", "frac"), "parity": g(c2, "parity_vs_oracle_first_and_last_image")} if "error" not in c2 else c2
    lay = extra.get("jpeg_layouts", {})
    out["layouts"] = {k: {"frac": g(v, "roofline", "frac"), "of_pattern": g(v, "roofline", "frac_of_pattern"), "parity": g(v, "parity_vs_oracle_first_image")}
                      for k, v in lay.items() if isinstance(v, dict)} if "error" not in lay else lay
    c4 = extra.get("c4", {})
    if "error" in c4:
        out["c4"] = c4
    else:
        sw = g(c4, "batch_sweep") or {}
        out["c4"] = {"frames16_random": {"value": g(c4, "value"), "ms": g(c4, "chain_ms"), "parity": g(c4, "parity_vs_reference_first_and_last_frame")},
                     "frames16_encoder": {"value": g(c4, "encoder_stream", "value"), "ms": g(c4, "encoder_stream", "chain_ms"),
                                          "parity": g(c4, "encoder_stream", "parity_vs_reference_whole_file_decode")},
                     "sweep_encoder": {str(r["frames"]): r["value"] for r in sw.get("encoder", [])},
                     "sweep_random": {str(r["frames"]): r["value"] for r in sw.get("random", [])},
                     "sweep_parity": [r.get("parity_first_and_last_frame") for src in ("encoder", "random") for r in sw.get(src, []) if "parity_first_and_last_frame" in r],
                     "fused_kernel": g(c4, "fused_roofline"),
                     "cpu_1_core": g(c4, "cpu_baseline", "value"), "cpu_all_cores": g(c4, "cpu_baseline_all_cores", "value"), "cpu_cores": g(c4, "cpu_baseline_all_cores", "cores")}
    c5 = extra.get("c5", {})
    if "error" in c5:
        out["c5"] = c5
    else:
        rows = g(c5, "grid", "rows") or []
        def planned(p):      # taken by the device planner with coding-tree wavefront tickets at the 64x64 window
            return bool(p and p.get("taken") and p.get("wavefront_tickets") and p.get("window_log2") == 6)
        out["c5"] = {"one_8k_picture": {"value": g(c5, "value"), "ms": g(c5, "chain_ms"), "intra_ms": g(c5, "stages", "intra_recon", "ms"), "parity": g(c5, "parity_vs_reference_sample")},
                     "one_8k_picture_reference_order": {"value": g(c5, "reference_order", "value"), "ms": g(c5, "reference_order", "chain_ms"), "intra_ms": g(c5, "reference_order", "intra_recon_ms"),
                                                        "same_pixels": g(c5, "reference_order", "same_pixels"), "wavefront_64": planned(g(c5, "reference_order", "plan")),
                                                        "sorted_by_plane": g(c5, "reference_order", "plan", "sorted_by_plane")},
                     "one_8k_picture_pipelined": {"value": g(c5, "pipelined", "value"), "ms": g(c5, "pipelined", "chain_ms"), "same_pixels": g(c5, "pipelined", "same_pixels")},
                     "grid_135_tiles": {str(r["pictures"]): r["value"] for r in rows},
                     "grid_135_tiles_reference_order": {str(r["pictures"]): g(r, "reference_order", "value") for r in rows},
                     "grid_135_tiles_unpipelined": {str(r["pictures"]): g(r, "one_call_unpipelined", "value") for r in rows},
                     "grid_135_tiles_three_calls": {str(r["pictures"]): g(r, "pipelined_three_calls", "value") for r in rows},
                     "grid_reference_order_ok": [bool(g(r, "reference_order", "same_pixels") and planned(g(r, "reference_order", "plan"))) for r in rows],
                     "grid_host_enqueue_ms": {str(r["pictures"]): r["intra_host_enqueue_ms"] for r in rows},
                     "grid_parity": [r.get("parity_first_and_last_tile_vs_reference") for r in rows if "parity_first_and_last_tile_vs_reference" in r],
                     "cpu_1_core": g(c5, "cpu_baseline", "value"), "cpu_all_cores": g(c5, "cpu_baseline_all_cores", "value"), "cpu_cores": g(c5, "cpu_baseline_all_cores", "cores")}
    f1 = extra.get("f1", {})
    if f1:
        out["f1"] = f1 if "error" in f1 else {k: {"value": g(v, "value"), "ms": g(v, "files_to_device_pixels_ms") or g(v, "ms_per_call_aggregate"), "entropy_gpu": g(v, "entropy_batch_gpu", "value"),
                                                   "device_pipeline_ms": g(v, "entropy_batch_gpu", "device_pipeline_ms") or g(v, "device_pipeline_ms"), "recon_ms": g(v, "reconstruction_ms"),
                                                   "parity": g(v, "parity_vs_reference_whole_file_decode"), "cpu_1_core": g(v, "cpu_baseline", "value"),
                                                   "cpu_all_cores": g(v, "cpu_baseline_all_cores", "value")} for k, v in (f1.get("files") or {}).items()}
    sk = extra.get("stage_kernels", {})
    out["stage_kernels"] = {k: g(v, "frac") for k, v in sk.items() if isinstance(v, dict)} if "error" not in sk else sk
    if isinstance(out["stage_kernels"], dict) and isinstance(sk.get("vp8_residual_256x1080p"), dict):
        out["stage_kernels"]["vp8_residual_of_pattern"] = sk["vp8_residual_256x1080p"].get("frac_of_pattern")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--images", type=int, default=0, help="override the number of images (per batch if strong, per GPU if weak)")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="weak (default): the configuration's batch per GPU (independent images, no data-path collective); strong: that batch shared out over the GPUs (BASELINE config 3 read literally: 256 images in total)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the C2 / C4 / C5 measurements of `extra`")
    ap.add_argument("--extras", default="", help="comma-separated subset of c2,jpeg_layouts,c4,c5,f1,stage_kernels to measure (default: all)")
    ap.add_argument("--extra-file", default="", help="also write the result line WITH the verbose `extra` object (every stage, roofline and sample description) to this file")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"--gpus {a.gpus} under a launcher with WORLD_SIZE={world}: start it as `python3 bench.py --gpus N` (it spawns its own ranks) "
                 "or as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    # Rehearsal of the N > 1 path on a one-GPU box (FFHIP_BENCH_REHEARSE=1): every rank on cuda:0, gloo instead of
    # RCCL (which refuses two ranks on one device).  Exercises sharding, barriers, the batch close and the
    # rank-0-only legs; its numbers mean nothing and the JSON line says so.
    rehearse = os.environ.get("FFHIP_BENCH_REHEARSE") == "1" and world > 1
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm: bootstrap and barriers only
    L = capi.require_device(local)                       # raises without gfx950: no CPU fallback

    cols, rows, batch_images, desc = WORKLOADS[a.workload]
    if a.images:
        batch_images = a.images
    total_images = batch_images if a.scaling == "strong" else batch_images * world
    first, last = shard.shard_range(total_images, rank, world)
    n = last - first
    geom = capi.jpeg_geom(cols, rows)
    H, W = geom.height, geom.width
    batch = shard.Batch(rank, world, dev)                # the C-side communicator for the batch close (RCCL from C when N > 1)

    t_y, t_u, t_v, q = gen_batch_on_device(dev, cols, rows, max(n, 1), seed=first)
    t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
    pitch, stride = W * 4, W * 4 * H
    out = torch.empty(max(n, 1) * stride, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ops.jpeg_recon_batch(geom, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0,
                             out.data_ptr(), pitch, stride, None, 0, stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # parity spot check on this rank's first AND last image, before warmup (outside the timed region;
    # the CPU-side check idles the GPU, so it must not sit between warmup and timing), and the per-image
    # checksums the rank vouches for its output with in the batch close
    parity = None
    step()
    sums = torch.zeros(max(n, 1), dtype=torch.int64, device=dev)
    capi.check(L.ffhip_bgra_checksum(out.data_ptr(), pitch, stride, W, H, n, sums.data_ptr(), stream), "ffhip_bgra_checksum")
    torch.cuda.synchronize()
    checksum = int(sums[:n].sum().item()) & 0xFFFFFFFFFFFFFFFF
    if rank == 0:
        O = oracle_lib()
        mcus = cols * rows
        parity = True
        for i in sorted({0, n - 1}):
            exp = O.oracle_jpeg_recon(O.make_geom(cols, rows), t_y[i * mcus * 256:(i + 1) * mcus * 256].cpu().numpy(),
                                      t_u[i * mcus * 64:(i + 1) * mcus * 64].cpu().numpy(), t_v[i * mcus * 64:(i + 1) * mcus * 64].cpu().numpy(), q)[0]
            got = out[i * stride:(i + 1) * stride].cpu().numpy().reshape(H, W, 4)
            words = exp.reshape(-1).view(np.uint32).astype(np.uint64)
            want = int((words * ((np.arange(words.size, dtype=np.uint64) & np.uint64(0xFFFF)) + np.uint64(1))).sum(dtype=np.uint64))
            parity = parity and bool(np.array_equal(got, exp)) and (int(sums[i].item()) & 0xFFFFFFFFFFFFFFFF) == want
    batch.close(first, n, 0, checksum, stream)       # warm the collective / small-copy path too

    ev0, ev1 = L.ffhip_event_create(), L.ffhip_event_create()
    barrier()
    for _ in range(a.warmup):      # W untimed warmup steps, immediately before the timed K
        step()
    barrier()
    t0 = time.perf_counter()
    capi.check(L.ffhip_event_record(ev0, stream))
    for _ in range(a.steps):
        step()
    capi.check(L.ffhip_event_record(ev1, stream))
    records = batch.close(first, n, 0, checksum, stream)     # closes the batch: ncclAllGather of the records, from C
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = L.ffhip_event_elapsed_ms(ev0, ev1) / max(a.steps, 1)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    complete = shard.batch_complete(records, total_images)

    # copy-kernel calibration of the achievable HBM rate, same process, same buffers
    copy_gbs = None
    if rank == 0:
        nbytes = min(out.numel(), 4 << 30) // 2 // 16 * 16
        src, dst = out[:nbytes], out[nbytes:2 * nbytes]
        for _ in range(2):
            capi.check(L.ffhip_copy_calibrate(dst.data_ptr(), src.data_ptr(), nbytes, stream))
        capi.check(L.ffhip_event_record(ev0, stream))
        for _ in range(5):
            capi.check(L.ffhip_copy_calibrate(dst.data_ptr(), src.data_ptr(), nbytes, stream))
        capi.check(L.ffhip_event_record(ev1, stream))
        capi.check(L.ffhip_stream_sync(stream))
        copy_gbs = 2 * nbytes * 5 / (L.ffhip_event_elapsed_ms(ev0, ev1) * 1e-3) / 1e9

    # the same launch into a buffer of the pitch the library recommends to callers that own their output (ffhip_bgra_layout: + 1 KiB per row);
    # reported next to the headline, which stays at the reference's pitch
    rec_pitch = None
    if rank == 0 and world == 1:
        try:
            rp, rs = C.c_int64(), C.c_int64()
            capi.check(L.ffhip_bgra_layout(C.byref(geom), C.byref(rp), C.byref(rs)), "ffhip_bgra_layout")
            out2 = torch.empty(max(n, 1) * rs.value, dtype=torch.uint8, device=dev)
            step2 = lambda: ops.jpeg_recon_batch(geom, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0, out2.data_ptr(), rp.value, rs.value, None, 0, stream)
            for _ in range(2):
                step2()
            capi.check(L.ffhip_event_record(ev0, stream))
            for _ in range(10):
                step2()
            capi.check(L.ffhip_event_record(ev1, stream))
            capi.check(L.ffhip_stream_sync(stream))
            ms2 = L.ffhip_event_elapsed_ms(ev0, ev1) / 10
            same = bool(torch.equal(out2.view(n, H, rp.value)[0, :, :W * 4], out.view(n, H, pitch)[0]) and torch.equal(out2.view(n, H, rp.value)[n - 1, :, :W * 4], out.view(n, H, pitch)[n - 1]))
            rec_pitch = {"pitch": rp.value, "value": round(n * H * W / ms2 / 1e3, 1), "frac": round(BYTES_PER_PIXEL * n * H * W / ms2 / 1e6 / HBM_PEAK_GBS, 4),
                         "kernel_ms": round(ms2, 4), "same_pixels_as_headline": same}
            del out2
        except Exception as e:
            rec_pitch = {"error": f"{type(e).__name__}: {e}"}

    # the fused kernel's own access pattern with the arithmetic taken out (ffhip_jpeg_pattern_calibrate: same grid, loads, store addresses) on the
    # HEADLINE's buffers -- the ceiling this placement of them allows; `out` holds meaningless bytes afterwards (nothing below reads it)
    pattern_gbs = None
    if rank == 0 and world == 1:
        try:
            cal = lambda: capi.check(L.ffhip_jpeg_pattern_calibrate(C.byref(geom), n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0,
                                                                      out.data_ptr(), pitch, stride, stream), "ffhip_jpeg_pattern_calibrate")
            for _ in range(2):
                cal()
            capi.check(L.ffhip_event_record(ev0, stream))
            for _ in range(10):
                cal()
            capi.check(L.ffhip_event_record(ev1, stream))
            capi.check(L.ffhip_stream_sync(stream))
            pattern_gbs = BYTES_PER_PIXEL * n * H * W / (L.ffhip_event_elapsed_ms(ev0, ev1) / 10 * 1e-3) / 1e9
        except Exception as e:
            pattern_gbs = None

    if rank == 0:
        px_per_launch = n * H * W
        achieved = BYTES_PER_PIXEL * px_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = pmc_traffic(a.workload) if (a.scaling == "weak" or world == 1) and n == WORKLOADS[a.workload][2] else None
        line = {
            "metric": "Mpixels/s decoded (dequant+IDCT+YUV->BGRA), 4K JPEG batch",
            "value": round(total_images * H * W * a.steps / dt / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / max(a.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "int16 -> int32 -> u8",
            "data": "synthetic (SURVEY 8d)",
            "config": {"workload": desc + (" in total" if a.scaling == "strong" else " per GPU"), "images_total": total_images,
                       "images_this_rank": n, "coded_size": f"{W}x{H}",
                       "subsampling": "4:2:0", "parallelism": f"image ranges over {world} GPU(s), no data-path collective",
                       "batch_close": {"none": "one GPU: stream sync + own record", "rccl": "ncclAllGather of 32-byte records from C",
                                       "torch": "records through torch.distributed"}[batch.transport],
                       "batch_complete": complete, "parity_vs_oracle_first_and_last_image": parity},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": None if traffic is None else "profiles/latest_pmc.json",
                         "kernel": "k_jpeg420_fused", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(BYTES_PER_PIXEL * px_per_launch),
                         "copy_kernel_GBps": None if copy_gbs is None else round(copy_gbs, 1),
                         "pattern_GBps": None if pattern_gbs is None else round(pattern_gbs, 1),
                         "frac_of_pattern": None if pattern_gbs is None else round(achieved / pattern_gbs, 4),
                         "at_recommended_pitch": rec_pitch},
        }
        if rehearse:
            line["config"]["rehearsal"] = "all ranks on one GPU over gloo: exercises the N > 1 control path only, the value is meaningless"
        if not a.no_cpu and world == 1:   # the CPU leg is timed on rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(cols, rows, t_y, t_u, t_v, q)
        if world == 1 and not a.no_extra:
            del t_y, t_u, t_v, out
            torch.cuda.empty_cache()
            T = Timer(L, stream)
            extra = {}
            for key, fn in (("c2", lambda: extra_c2(L, dev, stream, T)), ("jpeg_layouts", lambda: extra_layouts(L, dev, stream, T)),
                            ("c4", lambda: extra_c4(L, dev, stream, T, not a.no_cpu)), ("c5", lambda: extra_c5(L, dev, stream, T, not a.no_cpu)),
                            ("f1", lambda: extra_f1(L, dev, stream, T, not a.no_cpu)),
                            ("stage_kernels", lambda: extra_stage_kernels(L, dev, stream, T))):
                if a.extras and key not in a.extras.split(","):
                    continue
                try:
                    extra[key] = fn()
                except Exception as e:   # an extra must never take the headline line with it
                    extra[key] = {"error": f"{type(e).__name__}: {e}"}
            line["configs"] = compact_configs(extra)
            # the verbose record goes to stderr (one line, in FRONT of the result line) and, with --extra-file, to a file: the driver keeps the
            # last 8 KB of the output, and the one line it must find there whole is the result line below
            blob = json.dumps({"extra": extra})
            sys.stderr.write(blob + "\n")
            sys.stderr.flush()
            if a.extra_file:
                with open(a.extra_file, "w") as fh:
                    json.dump({**line, "extra": extra}, fh)
        print(json.dumps(line), flush=True)
    batch.destroy()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
