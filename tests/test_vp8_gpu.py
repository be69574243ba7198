"""GPU: batched VP8 residual stage against goldens and the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import ops, synth

pytestmark = pytest.mark.gpu


def oracle_residual(lv, info, q):
    out = np.zeros((lv.shape[0], 384), np.int16)
    F = O.ffo()
    for i in range(lv.shape[0]):
        F.ffo_vp8_residual_mb(np.ascontiguousarray(lv[i]).reshape(-1), info[i], int(info[i, 25]),
                              np.ascontiguousarray(q[info[i, 26], :6]), out[i])
    return out


def test_golden_macroblocks(golden):
    g = golden("vp8_mbs.npz")
    for tag in ("syn", "adv"):
        got = ops.vp8_residual_batch(g[f"{tag}_levels"], g[f"{tag}_info"], g["quant"])
        assert np.array_equal(got, g[f"{tag}_residual"]), tag


def test_golden_residual_block_driven(golden):
    """what the reference's vp8_decode_residual_block (webp.c:1125-1199) itself wrote while parsing synthetic streams"""
    g = golden("vp8_residual_driven.npz")
    for regime in ("random", "sparse", "dense"):
        got = ops.vp8_residual_batch(g[f"{regime}_levels"], g[f"{regime}_info"], g[f"{regime}_quant"])
        assert np.array_equal(got, g[f"{regime}_residual"]), regime


@pytest.mark.parametrize("n,adv", [(1, False), (7, False), (8, True), (9, True), (8160, False), (1000, True)])
def test_vs_oracle(n, adv):
    """ragged counts (partial workgroups), one 1080p frame worth of macroblocks, full-range levels"""
    lv, info = synth.vp8_macroblocks(n, seed=n, adversarial=adv)
    q = synth.vp8_quant(seed=n)
    assert np.array_equal(ops.vp8_residual_batch(lv, info, q), oracle_residual(lv, info, q))


def test_all_skip_and_all_y2_extremes():
    lv, info = synth.vp8_macroblocks(64, seed=5)
    q = synth.vp8_quant()
    for has in (0, 1):
        info[:, 25] = has
        for nzv in (0, 1, 2, 16):
            info[:, :25] = nzv
            assert np.array_equal(ops.vp8_residual_batch(lv, info, q), oracle_residual(lv, info, q)), (has, nzv)
