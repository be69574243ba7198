#!/usr/bin/env python3
"""K independent 512x512 HEVC pictures (the tiles of a HEIF grid) side by side in one plane set, reconstructed by ONE
ffhip_hevc_intra_recon call: the stage is bound by one wave's latency along each picture's dependency chain, so
independent tiles cost (almost) nothing extra until the machine fills."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ffpic_amd import capi, synth

dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
T = 512
tus0, res0 = synth.hevc_intra_tus(T, T, seed=3)
out = {"tile": [T, T], "tus_per_tile": int(len(tus0))}
for K in (1, 4, 16, 48, 96):
    tus = np.concatenate([tus0.copy() for _ in range(K)])
    for i in range(K):
        sl = slice(i * len(tus0), (i + 1) * len(tus0))
        tus["x"][sl] += np.where(tus0["cidx"] == 0, T * i, T // 2 * i).astype(np.uint16)
        tus["res_offset"][sl] += len(res0) * i
    res = np.tile(res0, K)
    W, H = T * K, T
    dt = torch.from_numpy(tus.view(np.uint8).copy()).to(dev); dr = torch.from_numpy(res).to(dev)
    py = torch.zeros((H, W), dtype=torch.int16, device=dev); pu = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=dev); pv = torch.zeros_like(pu)
    def run():
        capi.check(L.ffhip_hevc_intra_recon(tus.ctypes.data, dt.data_ptr(), len(tus), dr.data_ptr(), py.data_ptr(), pu.data_ptr(), pv.data_ptr(), W, H, W, W // 2, H // 2, W // 2, 8, 8, st))
        capi.check(L.ffhip_stream_sync(st))
    run()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
    if K == 4:  # parity of the packed layout against the oracle
        import oracle_lib as O
        exp = O.oracle_hevc_intra(tus, res, W, H, True, 8, 8)
        assert np.array_equal(py.cpu().numpy(), exp[0]) and np.array_equal(pu.cpu().numpy(), exp[1])
    out[f"tiles_{K}"] = {"wall_ms": round(best * 1e3, 2), "Mpx/s": round(K * T * T / best / 1e6, 1)}
print(json.dumps(out, indent=1))
