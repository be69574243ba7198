"""GPU: batched HEVC scaling + inverse transforms against the
golden vectors (produced by the reference's scale_transform_coefficients /
transform_scaled_coeffients / idct_4x4_hevc) and the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops

pytestmark = pytest.mark.gpu


def oracle_tus(n, level, info, bitdepth, epp, scaling):
    out = np.zeros_like(level)
    F = O.ffo()
    for i in range(level.shape[0]):
        sf = None if scaling is None else scaling[info[i, 2]].ctypes.data_as(C.c_void_p)
        F.ffo_hevc_residual_tu(np.ascontiguousarray(level[i]), out[i], n, int(info[i, 0]), int(info[i, 1]), bitdepth,
                               int(epp), sf)
    return out


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_golden_scale_and_dct(golden, n):
    g = golden("hevc_transform.npz")
    lv = g[f"level_{n}"]
    for bd in (8, 10):
        for qp in (0, 22, 37, 51):
            info = np.zeros((lv.shape[0], 4), np.uint8)
            info[:, 0] = qp
            got = ops.hevc_residual_batch(n, lv, info, bitdepth=bd)
            assert np.array_equal(got, g[f"r_{n}_bd{bd}_qp{qp}"]), (n, bd, qp)


def test_golden_dst4(golden):
    """idct_4x4_hevc goldens: feed d directly by using bypass-free identity scaling is not
    possible, so compare through the oracle (itself pinned on these goldens) on levels."""
    g = golden("hevc_dst4.npz")
    lv = g["coef"]
    for bd, epp in ((8, 0), (10, 0), (8, 1), (10, 1)):
        for qp in (4, 30):
            info = np.zeros((lv.shape[0], 4), np.uint8)
            info[:, 0], info[:, 1] = qp, 1
            got = ops.hevc_residual_batch(4, lv, info, bitdepth=bd, epp=bool(epp))
            assert np.array_equal(got, oracle_tus(4, lv, info, bd, epp, None)), (bd, epp, qp)


def test_golden_scale_and_transform_glue(golden):
    """a12, the golden twin of test_mixed_flags_vs_oracle: residuals the reference's own scale_and_transform
    (hevc.c:4172-4251) produced on its bypass / transform-skip / rotation branches, per TU size one mixed batch"""
    from test_oracle_golden import glue_cases
    g = golden("hevc_scale_and_transform.npz")
    groups = {}
    for case in glue_cases(g):
        key, kind, n, cidx, bd, epp, flags, qp, sf = case
        groups.setdefault((n, bd, epp), []).append(case)
    checked = 0
    for (n, bd, epp), cases in groups.items():
        lv0 = g[f"level_{n}"]
        nb = lv0.shape[0]
        scaling = np.ones((6, n * n), np.uint8)
        scaling[1] = g[f"sfactor_{n}"]                     # matrixId 1 = the golden list, 0 = a list of ones is NOT flat 16:
        for with_sf in (False, True):                      # so batches are split by "has a scaling list"
            sel = [c for c in cases if (c[8] is not None) == with_sf]
            if not sel:
                continue
            lv = np.concatenate([lv0] * len(sel))
            info = np.zeros((len(lv), 4), np.uint8)
            exp = np.concatenate([g[c[0]] for c in sel])
            for k, c in enumerate(sel):
                info[k * nb:(k + 1) * nb, 0] = c[7]
                info[k * nb:(k + 1) * nb, 1] = c[6]
                info[k * nb:(k + 1) * nb, 2] = 1
            got = ops.hevc_residual_batch(n, lv, info, bitdepth=bd, epp=bool(epp), scaling=scaling if with_sf else None)
            bad = np.nonzero((got != exp).any(axis=1))[0]
            assert bad.size == 0, (n, bd, epp, with_sf, [sel[b // nb][0] for b in bad[:4]])
            checked += len(sel)
    assert checked == 6 * 7 * 4


@pytest.mark.parametrize("n,n_tu", [(4, 1), (4, 1000), (8, 37), (16, 9), (32, 5), (32, 64)])
def test_mixed_flags_vs_oracle(n, n_tu):
    rng = np.random.default_rng(n * 1000 + n_tu)
    lv = np.rint(rng.laplace(0, 10, size=(n_tu, n * n))).astype(np.int16)
    lv[::7] = rng.integers(-32768, 32768, size=(len(lv[::7]), n * n)).astype(np.int16)
    info = np.zeros((n_tu, 4), np.uint8)
    info[:, 0] = rng.integers(0, 52, size=n_tu)
    flags = rng.choice([0, 0, 0, 2, 4, 2 | 8, 4 | 8] + ([1, 1] if n == 4 else []), size=n_tu)
    if n != 4:
        flags = flags & ~8          # rotateCoeffs exists for 4x4 only (hevc.c:4199-4203)
    info[:, 1] = flags
    info[:, 2] = rng.integers(0, 6, size=n_tu)
    scaling = rng.integers(1, 256, size=(6, n * n)).astype(np.uint8)
    for bd, epp, sc in ((8, False, None), (10, False, scaling), (12, True, scaling)):
        got = ops.hevc_residual_batch(n, lv, info, bitdepth=bd, epp=epp, scaling=sc)
        assert np.array_equal(got, oracle_tus(n, lv, info, bd, epp, sc)), (bd, epp, sc is not None)


@pytest.mark.parametrize("env", [{"FFHIP_HEVC_RES_ITERS": "3"}, {"FFHIP_HEVC_RES32": "dot", "FFHIP_HEVC_RES16": "dot", "FFHIP_HEVC_RES8": "dot", "FFHIP_HEVC_RES4": "rows"},
                                 {"FFHIP_HEVC_RES32": "dot", "FFHIP_HEVC_RES16": "dot", "FFHIP_HEVC_RES8": "dot", "FFHIP_HEVC_RES4": "rows", "FFHIP_HEVC_RES_ITERS": "2"}])
@pytest.mark.parametrize("n,n_tu", [(4, 777), (8, 203), (16, 101), (32, 51)])
def test_kernel_variants_agree(env, n, n_tu, monkeypatch):
    """Several batches per wave (with the ragged tail inside a wave's run), and the butterfly / rows kernels the
    32x32 and 4x4 sizes no longer use by default."""
    for k, v in env.items():
        monkeypatch.setenv(k, v); capi.reload_env()
    rng = np.random.default_rng(n + n_tu)
    lv = rng.integers(-32768, 32768, size=(n_tu, n * n)).astype(np.int16)
    lv[1::2] = np.rint(rng.laplace(0, 6, size=lv[1::2].shape)).astype(np.int16)
    info = np.zeros((n_tu, 4), np.uint8)
    info[:, 0] = rng.integers(0, 52, size=n_tu)
    info[:, 1] = rng.choice([0, 0, 0, 0, 2, 4, 2 | 8, 4 | 8] + ([1, 1] if n == 4 else []), size=n_tu)
    if n != 4:
        info[:, 1] &= ~np.uint8(8)
    info[:, 2] = rng.integers(0, 6, size=n_tu)
    scaling = rng.integers(1, 256, size=(6, n * n)).astype(np.uint8)
    for bd, epp, sc in ((8, False, scaling), (12, True, None)):
        got = ops.hevc_residual_batch(n, lv, info, bitdepth=bd, epp=epp, scaling=sc)
        assert np.array_equal(got, oracle_tus(n, lv, info, bd, epp, sc)), (bd, epp)


def test_extreme_coefficients_32():
    """The 32x32 matrix-core path splits int16 data into bytes: saturated inputs of both signs, every qP."""
    n = 32
    pats = [np.full(n * n, 32767), np.full(n * n, -32768), np.where(np.arange(n * n) % 2, 32767, -32768),
            np.where((np.arange(n * n) // n) % 2, -32768, 32767), np.eye(n).ravel() * 32767, np.full(n * n, 255), np.full(n * n, -256),
            np.full(n * n, 128), np.full(n * n, -129)]
    lv = np.stack(pats).astype(np.int16)
    lv = np.concatenate([lv] * 6)
    info = np.zeros((lv.shape[0], 4), np.uint8)
    info[:, 0] = np.repeat(np.array([0, 5, 17, 30, 44, 51], np.uint8), len(pats))
    for bd, epp in ((8, False), (10, False), (16, True)):
        got = ops.hevc_residual_batch(n, lv, info, bitdepth=bd, epp=epp)
        assert np.array_equal(got, oracle_tus(n, lv, info, bd, epp, None)), (bd, epp)


@pytest.mark.parametrize("n", [4, 8, 16, 32])
def test_full_8k_plane_by_pattern_gather(n):
    """BASELINE config 5 size: every TU of one 7680x4320 luma plane.  The oracle transforms K distinct TUs (levels, qP,
    flags, matrixId all differ); the plane is those K patterns drawn at random, so the expected output is a gather of
    K oracle results -- exact parity at full size for the price of K oracle calls.  Runs several batches per wave,
    the ragged last workgroup and every kernel of the default dispatch."""
    K = 61
    rng = np.random.default_rng(8000 + n)
    pat = np.rint(rng.laplace(0, 12, size=(K, n * n))).astype(np.int16)
    pat[::9] = rng.integers(-32768, 32768, size=(len(pat[::9]), n * n)).astype(np.int16)
    pinfo = np.zeros((K, 4), np.uint8)
    pinfo[:, 0] = rng.integers(0, 52, size=K)
    pinfo[:, 1] = rng.choice([0, 0, 0, 0, 0, 2, 4] + ([1, 1, 1, 2 | 8] if n == 4 else []), size=K)
    pinfo[:, 2] = rng.integers(0, 6, size=K)
    scaling = rng.integers(1, 256, size=(6, n * n)).astype(np.uint8)
    n_tu = (7680 // n) * (4320 // n) - 3      # not a multiple of anything
    pick = rng.integers(0, K, size=n_tu)
    lv, info = pat[pick], pinfo[pick]
    for bd, epp, sc in ((8, False, None), (10, False, scaling)):
        want = oracle_tus(n, pat, pinfo, bd, epp, sc)[pick]
        got = ops.hevc_residual_batch(n, lv, info, bitdepth=bd, epp=epp, scaling=sc)
        assert np.array_equal(got, want), (n, bd)
