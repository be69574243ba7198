"""CPU, build container only: the oracle against the reference itself
(oracle/_ref/libffpic_ref.so, compiled from /root/reference by oracle/Makefile) on
fresh random inputs -- wider than the committed goldens.  Skipped where neither the
reference sources nor a prebuilt oracle/_ref exist."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import synth

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="no /root/reference and no prebuilt oracle/_ref")


@pytest.fixture(scope="module")
def libs():
    return O.ffo(), O.ref()


def test_idct8x8_and_dequant_random(libs):
    F, R = libs
    rng = np.random.default_rng(11)
    blocks = np.concatenate([synth.adversarial_blocks(rng, 2048), synth._blocks(rng, 2048, synth.quant_tables()[0])])
    q = rng.integers(1, 65536, size=64).astype(np.uint16)
    for b in blocks:
        a, r = b.copy(), b.copy()
        F.ffo_idct_8x8_16(a)
        R.ref_idct_8x8_16(r)
        assert np.array_equal(a, r)
        da, dr = np.zeros(64, np.int16), np.zeros(64, np.int16)
        F.ffo_jpeg_dequant(da, b.copy(), q, 63)
        R.ref_jpeg_dequant(dr, b.copy(), q.copy(), 63)
        assert np.array_equal(da, dr)


def test_vp8_random(libs):
    F, R = libs
    rng = np.random.default_rng(12)
    for b in rng.integers(-32768, 32768, size=(4096, 16)).astype(np.int16):
        a, r = b.copy(), b.copy()
        F.ffo_vp8_idct_4x4(a)
        R.ref_vp8_idct_4x4(r)
        assert np.array_equal(a, r)
        wa, wr = np.zeros(256, np.int16), np.zeros(256, np.int16)
        F.ffo_vp8_iwht_long(b.copy(), wa)
        R.ref_vp8_iwht_long(b.copy(), wr)
        assert np.array_equal(wa, wr)


@pytest.mark.parametrize("regime", ["random", "sparse", "dense"])
def test_vp8_residual_block_driven(libs, regime):
    """ffo_vp8_residual_mb against vp8_decode_residual_block itself (webp.c:1125-1199) on fresh synthetic streams"""
    F, _ = libs
    lv, info, q, exp = O.ref_vp8_driven(400, seed=77, regime=regime)
    for i in range(lv.shape[0]):
        out = np.zeros(384, np.int16)
        F.ffo_vp8_residual_mb(np.ascontiguousarray(lv[i]).reshape(-1), info[i], int(info[i, 25]),
                              np.ascontiguousarray(q[info[i, 26], :6]), out)
        assert np.array_equal(out, exp[i]), i


def test_hevc_random(libs):
    F, R = libs
    rng = np.random.default_rng(13)
    for b in rng.integers(-32768, 32768, size=(1024, 16)).astype(np.int16):
        for bd, epp in ((8, 0), (10, 0), (12, 1)):
            a, r = np.zeros(16, np.int16), np.zeros(16, np.int16)
            F.ffo_hevc_idct_4x4_dst(b.copy(), a, bd, epp)
            R.idct_4x4_hevc(b.copy(), r, bd, bool(epp))
            assert np.array_equal(a, r)
    for n in (4, 8, 16, 32):
        for _ in range(24):
            lv = rng.integers(-32768, 32768, size=n * n).astype(np.int16) if rng.random() < 0.3 else \
                np.rint(rng.laplace(0, 8, size=n * n)).astype(np.int16)
            qp, bd = int(rng.integers(0, 52)), int(rng.choice([8, 10]))
            da, dr = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16)
            F.ffo_hevc_scale(lv.copy(), da, n, qp, bd, 0, None)
            R.ref_hevc_scale(lv.copy(), dr, n, qp, bd, 0, None, 0)
            assert np.array_equal(da, dr)
            ra, rr = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16)
            F.ffo_hevc_transform(da.copy(), ra, n, 0, bd, 0)
            R.ref_hevc_transform(dr.copy(), rr, n, 0, bd, 0)
            assert np.array_equal(ra, rr)


def test_hevc_scale_and_transform_every_branch(libs):
    """ffo_hevc_residual_tu against the reference's scale_and_transform (hevc.c:4172-4251) called as a whole: random
    bypass / transform-skip / rotation flags, scaling lists, bit depths, extended precision, luma and chroma (the qP
    the reference derives in 8.6.1 is handed to the restatement)"""
    import ctypes as C
    F, R = libs
    rng = np.random.default_rng(0xA12)
    seen = set()
    for n in (4, 8, 16, 32):
        for _ in range(150 if n < 32 else 60):
            lv = rng.integers(-32768, 32768, size=n * n).astype(np.int16) if rng.random() < 0.3 else \
                np.rint(rng.laplace(0, 8, size=n * n)).astype(np.int16)
            bd, epp = int(rng.choice([8, 10, 12])), int(rng.random() < 0.2)
            qp = int(rng.integers(0, 52 + 6 * (bd - 8)))
            cidx, cat = int(rng.integers(0, 3)), int(rng.choice([1, 3]))
            bypass, ts, rot = int(rng.random() < 0.25), int(rng.random() < 0.4), int(rng.random() < 0.5)
            sf = rng.integers(1, 256, size=n * n).astype(np.uint8) if rng.random() < 0.5 else None
            sfp = None if sf is None else sf.ctypes.data_as(C.c_void_p)
            r, a = np.zeros(n * n, np.int16), np.zeros(n * n, np.int16)
            qP = R.ref_hevc_scale_and_transform(lv.copy(), r, n, cidx, qp, bd, epp, bypass, ts, rot, cat, sfp)
            if cidx == 0:
                assert qP == qp
            flags = (1 if n == 4 and cidx == 0 else 0) | (2 if ts else 0) | (4 if bypass else 0) | (8 if rot and n == 4 else 0)
            F.ffo_hevc_residual_tu(lv.copy(), a, n, qP, flags, bd, epp, sfp)
            assert np.array_equal(a, r), (n, bd, epp, qp, cidx, cat, bypass, ts, rot, sf is not None)
            seen.add((bypass, ts, bool(rot and n == 4), sf is not None and n > 4))
    assert len(seen) == 12          # rotation exists at 4x4 only, the dropped scaling list above 4x4 only


@pytest.mark.parametrize("nc,h,v", [(3, 2, 2), (3, 1, 1), (3, 2, 1), (3, 1, 2), (1, 1, 1), (3, 4, 1), (3, 1, 4), (3, 3, 1),
                                    (3, 1, 3), (1, 2, 2), (1, 4, 1)])
def test_jpeg_grid_random(libs, nc, h, v):
    q = synth.quant_tables(60)
    g = O.make_geom(9, 5, nc, h, v)
    cy, cu, cv = synth.coef_batch(1, 9, 5, nc, h, v, quant=q, first=100)
    assert np.array_equal(O.oracle_jpeg_recon(g, cy, cu, cv, q)[0], O.ref_jpeg_recon(g, cy, cu, cv, q))


@pytest.mark.parametrize("seed,bd,bdc,c444", [(21, 8, 8, False), (22, 10, 10, False), (23, 8, 8, True), (24, 12, 9, True)])
def test_hevc_intra_random_tu_lists(libs, seed, bd, bdc, c444):
    """fresh TU lists (not the committed ones) through the reference's own neighbour processing,
    predictors, rdpcm, cross-component prediction and construct_pic vs the restatement"""
    from ffpic_amd import synth
    w, h = 128, 128
    tus, res = synth.hevc_intra_tus(w, h, seed, adversarial_masks=True, ccp=c444, chroma_444=c444)
    res = res.copy()
    res[::41] = 32767
    res[5::43] = -32768
    csub = 1 if c444 else 2
    exp = O.ref_hevc_intra(tus, res, w, h, True, bd, bdc, csub=csub)
    got = O.oracle_hevc_intra(tus, res, w, h, True, bd, bdc, csub=csub)
    for a, b, name in zip(got, exp, "YUV"):
        assert np.array_equal(a, b), name


def test_vp8_filter_params_sweep(libs):
    """f3: ffhip_vp8_filter_params (host C in the product) against the reference's calculate_filter_control_parameter
    (webp.c:1756-1803) over every level x sharpness x filter type, with and without segmentation (both feature modes)
    and loop-filter deltas, for 1, 2 and 4 partitions (the reference derives triples per PARTITION index, webp.c:1905-1915)"""
    import ctypes as C
    from ffpic_amd import capi
    _, R = libs
    L = capi.lib()
    rng = np.random.default_rng(3)
    cases = [(ft, lvl, sh, 0, 0, (0, 0, 0, 0), 0, 0, 0, 4) for ft in (0, 1) for lvl in range(64) for sh in range(8)]
    for _ in range(3000):
        cases.append((int(rng.integers(0, 2)), int(rng.integers(0, 64)), int(rng.integers(0, 8)), int(rng.integers(0, 2)), int(rng.integers(0, 2)),
                      tuple(int(x) for x in rng.integers(-63, 64, size=4)), int(rng.integers(0, 2)), int(rng.integers(-63, 64)),
                      int(rng.integers(-63, 64)), int(rng.choice([1, 2, 4]))))
    seen_types = set()
    for ft, lvl, sh, seg, fmode, lfu, adj, d0, d1, parts in cases:
        hdr = np.array([ft, lvl, sh, seg, fmode, *lfu, adj, d0, d1, parts], np.int32)
        exp = np.zeros(24, np.int32)
        R.ref_webp_filter_params(hdr, exp)
        h = capi.Vp8FilterHeader(ft, lvl, sh, seg, fmode, (C.c_int8 * 4)(*lfu), adj, d0, d1, parts)
        got = np.zeros(24, np.uint8)
        ftype = C.c_int(-1)
        assert L.ffhip_vp8_filter_params(C.byref(h), got.ctypes.data, C.byref(ftype)) == 0
        assert np.array_equal(got.astype(np.int32), exp), (ft, lvl, sh, seg, fmode, lfu, adj, d0, d1, parts)
        assert ftype.value == (0 if lvl == 0 else (1 if ft else 2))
        seen_types.add(ftype.value)
    assert seen_types == {0, 1, 2}
    bad = capi.Vp8FilterHeader(0, 64, 0, 0, 0, (C.c_int8 * 4)(), 0, 0, 0, 1)
    assert L.ffhip_vp8_filter_params(C.byref(bad), got.ctypes.data, C.byref(ftype)) == capi.FFHIP_EINVAL
    bad = capi.Vp8FilterHeader(0, 10, 0, 0, 0, (C.c_int8 * 4)(), 0, 0, 0, 3)
    assert L.ffhip_vp8_filter_params(C.byref(bad), got.ctypes.data, C.byref(ftype)) == capi.FFHIP_EINVAL
