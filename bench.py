#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the fused JPEG reconstruction (dequant + IDCT + YUV->BGRA)
on MI355X, the metric BASELINE.json names.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: per GPU, 256 synthetic
3840x2160 4:2:0 coefficient grids (BASELINE config 3 geometry) already resident in
HBM, reconstructed to BGRA in HBM by ONE launch of k_jpeg420_fused.  Images are
independent, so ranks own disjoint image ranges (weak scaling: per-GPU work fixed);
the only collective is a tiny all-gather of per-rank status records (RCCL) that
closes the batch.  Rank 0 prints one JSON line.

PyTorch is plumbing here: device memory, the stream handle and torch.distributed.
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

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ffpic_amd import capi, ops, shard, synth  # noqa: E402

WORKLOADS = {
    # name: (mcu_cols, mcu_rows, images per GPU, description)
    "c3": (240, 135, 256, "C3: 256 x 3840x2160 4:2:0 JPEG coefficient grids per GPU"),
    "c2": (120, 68, 1024, "C2: 1024 x 1920x1088 4:2:0 JPEG coefficient grids per GPU"),
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


def cpu_baseline(cols, rows, t_y, t_u, t_v, q, budget_s=20.0):
    """The reference's own C (oracle/_ref, compiled from /root/reference in the build
    container) -- or our bit-exact port when that .so did not travel -- timed on the
    host cores over a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
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
    corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, KiB units)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        if rec.get("workload") == workload:
            return rec.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--images", type=int, default=0, help="override images per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # Rehearsal of the N > 1 path on a one-GPU box (FFHIP_BENCH_REHEARSE=1): every rank on cuda:0, gloo instead of
    # RCCL (which refuses two ranks on one device).  Exercises sharding, barriers, the status gather and the
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
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
    L = capi.require_device(local)                       # raises without gfx950: no CPU fallback

    cols, rows, per_gpu, desc = WORKLOADS[a.workload]
    if a.images:
        per_gpu = a.images
    total_images = per_gpu * world
    first, last = shard.shard_range(total_images, rank, world)
    n = last - first
    geom = capi.jpeg_geom(cols, rows)
    H, W = geom.height, geom.width

    t_y, t_u, t_v, q = gen_batch_on_device(dev, cols, rows, n, seed=first)
    t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
    pitch, stride = W * 4, W * 4 * H
    out = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ops.jpeg_recon_batch(geom, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0,
                             out.data_ptr(), pitch, stride, None, 0, stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # parity spot check on this rank's first image, before warmup (outside the timed region;
    # the CPU-side check idles the GPU, so it must not sit between warmup and timing)
    parity = None
    step()
    torch.cuda.synchronize()
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        mcus = cols * rows
        exp = O.oracle_jpeg_recon(O.make_geom(cols, rows), t_y[:mcus * 256].cpu().numpy(),
                                  t_u[:mcus * 64].cpu().numpy(), t_v[:mcus * 64].cpu().numpy(), q)[0]
        parity = bool(np.array_equal(out[:stride].cpu().numpy().reshape(H, W, 4), exp))
    shard.gather_status(first, n, 0, device=dev)   # warm the collective / small-copy path too

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
    records = shard.gather_status(first, n, 0, device=dev)   # closes the batch (RCCL all-gather)
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
        copy_gbs = 2 * nbytes * 5 / (L.ffhip_event_elapsed_ms(ev0, ev1) * 1e-3) / 1e9

    if rank == 0:
        px_per_launch = n * H * W
        achieved = BYTES_PER_PIXEL * px_per_launch / (kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "Mpixels/s decoded (dequant+IDCT+YUV->BGRA), 4K JPEG batch",
            "value": round(total_images * H * W * a.steps / dt / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / max(a.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 coefficients -> int32 accumulate -> u8 BGRA (fp64 only for exact-integer G cases)",
            "data": "synthetic (device RNG; Annex-K q85 tables, Laplace AC / normal DC, SURVEY 8d)",
            "config": {"workload": desc, "images_per_gpu": per_gpu, "coded_size": f"{W}x{H}",
                       "subsampling": "4:2:0", "parallelism": f"images sharded over {world} GPU(s), no data-path collective",
                       "batch_complete": complete, "parity_vs_oracle_first_image": parity},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(a.workload),
                         "kernel": "k_jpeg420_fused", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": int(BYTES_PER_PIXEL * px_per_launch),
                         "copy_kernel_GBps": None if copy_gbs is None else round(copy_gbs, 1)},
        }
        if rehearse:
            line["config"]["rehearsal"] = "all ranks on one GPU over gloo: exercises the N > 1 control path only, the value is meaningless"
        if not a.no_cpu and world == 1:   # the CPU leg is timed on rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(cols, rows, t_y, t_u, t_v, q)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
