"""GPU: every FFHIP_* switch that selects another form of a product kernel, a launch shape or a diagnostics print gives the same
bytes as the shipped configuration.  A switch is a code path in the product: one that changes bytes or deadlocks is a bug
whether or not anyone sets it (round 4's FFHIP_HEVC_INTRA_WAVES=3 with four ticket counters was exactly that).  The switches
the other test files already drive (window, waves, shards, inline forms, planner choice, ...) are not repeated here."""
import os

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth
from test_color_gpu import oracle_420_8
from test_jpeg_gpu import gpu_recon
from test_oracle_golden import FILES

pytestmark = pytest.mark.gpu


@pytest.fixture
def switch(monkeypatch):
    """set library switches for the test and have them read again; undone (and read again) afterwards"""
    def set_(**kw):
        for k, v in kw.items():
            monkeypatch.setenv(k, str(v))
        capi.reload_env()
    yield set_
    monkeypatch.undo()
    capi.reload_env()


@pytest.mark.parametrize("env", [{"FFHIP_JPEG_VARIANT": v} for v in ("10", "11", "12", "13", "20", "21", "22", "23")] +
                         [{"FFHIP_JPEG_NO_XCD_REMAP": "1"}, {"FFHIP_JPEG_XCD_CHUNK_LOG2": "2"}, {"FFHIP_JPEG_XCD_CHUNK_LOG2": "5"},
                          {"FFHIP_JPEG_XCD_CHUNK_LOG2": "5", "FFHIP_JPEG_VARIANT": "21"}, {"FFHIP_JPEG_LDS_PAD": "16384"}, {"FFHIP_JPEG_LDS_PAD": "61440"}])
def test_jpeg_launch_shapes(env, switch):
    """quads per wave, store policy and the workgroup -> XCD mapping of k_jpeg420_fused: ragged and whole MCU counts, several images"""
    switch(**env)
    for cols, rows, n in ((13, 7, 3), (40, 30, 2), (120, 68, 1), (1, 1, 1)):
        geom = O.make_geom(cols, rows)
        q = synth.quant_tables(80)
        cy, cu, cv = synth.coef_batch(n, cols, rows, quant=q)
        exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n, n_threads=4)
        assert np.array_equal(gpu_recon(geom, n, cy, cu, cv, q), exp), (env, cols, rows, n)


@pytest.mark.parametrize("strips", ["1", "2"])
@pytest.mark.parametrize("nc,h,v", [(3, 1, 1), (3, 2, 1), (3, 1, 2), (1, 1, 1), (3, 4, 1), (3, 1, 4)])
def test_jpeg_strips_per_wave(nc, h, v, strips, switch):
    """FFHIP_JPEG_STRIPS=1 / 2: one or two strips' worth per wave of k_jpeg_fused_strip (h * v = 4 always has two) -- whole and ragged MCU counts (a strip of 16, 8
    or 4 MCUs cut anywhere), several images, per-image quantiser tables, and the exact-integer green branch via adversarial coefficients"""
    switch(FFHIP_JPEG_STRIPS=strips)
    for cols, rows, n in ((1, 1, 1), (7, 3, 2), (9, 2, 3), (16, 5, 1), (17, 4, 2), (33, 3, 1), (64, 9, 1)):
        geom = O.make_geom(cols, rows, nc, h, v)
        q = np.stack([synth.quant_tables(40 + 9 * i) for i in range(n)])
        cy, cu, cv = synth.coef_batch(n, cols, rows, nc, h, v, quant=q[0])
        exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n, n_threads=4)
        assert np.array_equal(gpu_recon(geom, n, cy, cu, cv, q), exp), (nc, h, v, strips, cols, rows, n)
    rng = np.random.default_rng(77)                                  # full-range coefficients: int16 wrap, clamps, exact-integer G
    cols, rows, n = 18, 3, 1
    geom = O.make_geom(cols, rows, nc, h, v)
    by = cols * rows * (h * v if nc == 3 else 1)
    cy = rng.integers(-32768, 32768, size=by * 64, dtype=np.int16)
    cu = rng.integers(-2048, 2048, size=cols * rows * 64, dtype=np.int16) if nc == 3 else None
    cv = rng.integers(-2048, 2048, size=cols * rows * 64, dtype=np.int16) if nc == 3 else None
    q = synth.quant_tables(95)
    assert np.array_equal(gpu_recon(geom, n, cy, cu, cv, q), O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n)), (nc, h, v, strips, "adversarial")


def test_vp8_residual_pattern_twin(switch):
    """FFHIP_VP8_RESIDUAL_PATTERN=1 (bench.py's stage_kernels.vp8_residual_of_pattern): k_vp8_residual's loads and stores without the arithmetic -- it writes the residual
    buffer (every macroblock's 768 bytes, nothing beyond) with meaningless words, and the next real call is exact again"""
    L = capi.require_device()
    n_mb = 999
    lv, info = synth.vp8_macroblocks(n_mb, seed=4)
    q = synth.vp8_quant()
    from test_vp8_gpu import oracle_residual
    exp = oracle_residual(lv, info, q)
    dl, di = ops.DeviceBuffer(np.ascontiguousarray(lv)), ops.DeviceBuffer(np.ascontiguousarray(info))
    dq = ops.DeviceBuffer(np.ascontiguousarray(q.astype(np.uint16)))
    out = ops.DeviceBuffer(nbytes=n_mb * 768 + 256)
    capi.check(L.ffhip_memset(out.ptr, 0xA5, n_mb * 768 + 256, None))
    switch(FFHIP_VP8_RESIDUAL_PATTERN=1)
    capi.check(L.ffhip_vp8_residual_batch(n_mb, dl.ptr, di.ptr, dq.ptr, out.ptr, None))
    capi.check(L.ffhip_stream_sync(None))
    raw = out.to_host((n_mb * 768 + 256,), np.uint8)
    assert (raw[n_mb * 768:] == 0xA5).all()
    assert (raw[:n_mb * 768].view(np.uint32) != 0xA5A5A5A5).mean() > 0.99
    switch(FFHIP_VP8_RESIDUAL_PATTERN="")
    os.environ.pop("FFHIP_VP8_RESIDUAL_PATTERN", None)
    capi.reload_env()
    capi.check(L.ffhip_vp8_residual_batch(n_mb, dl.ptr, di.ptr, dq.ptr, out.ptr, None))
    capi.check(L.ffhip_stream_sync(None))
    assert np.array_equal(out.to_host((n_mb, 384), np.int16), exp.reshape(n_mb, 384))


def test_color8_scalar_form(switch):
    """FFHIP_COLOR8_SCALAR=1: the one-pixel-per-lane form of the 8-bit planar converter instead of the packed one -- all 65 536 chroma pairs"""
    rng = np.random.default_rng(3)
    uu, vv = np.meshgrid(np.arange(256), np.arange(256))
    u, v = uu.astype(np.uint8)[None], vv.astype(np.uint8)[None]
    y = rng.integers(0, 256, size=(1, 512, 512)).astype(np.uint8)
    exp = oracle_420_8(y[0], u[0], v[0], 32, 32)
    assert np.array_equal(ops.yuv420_to_bgra(y, u, v, 32, 32)[0], exp)
    switch(FFHIP_COLOR8_SCALAR=1)
    assert np.array_equal(ops.yuv420_to_bgra(y, u, v, 32, 32)[0], exp)


def _vp8_case(c=21, r=13, n=3, seed=1500):
    modes = np.stack([synth.vp8_modes(c, r, seed=seed + i) for i in range(n)])
    modes.reshape(n, r, c, 20)[:, 1::2, 0, 0] = 3                   # every other row starts with the wrapped H_PRED
    resid = np.stack([synth.vp8_residual(c * r, seed=seed + 10 + i) for i in range(n)])
    return c, r, n, modes, resid, synth.vp8_filters(seed=seed % 97)


def test_vp8_frames_min_moves_the_form_switch(switch):
    """FFHIP_VP8_FRAMES_MIN: the batch size from which ffhip_vp8_decode_frames takes the frame kernel (half the compute units by default);
    ffhip_vp8_decode_frames_form reports the form, and both forms give the same BGRA"""
    from test_vp8_frames_gpu import oracle_chain
    L = capi.require_device()
    c, r, n, modes, resid, flt = _vp8_case()
    assert L.ffhip_vp8_decode_frames_form(n) == 0 and L.ffhip_vp8_decode_frames_form(4096) == 1
    rows_form = ops.vp8_decode_frames(c, r, modes, resid, 2, flt)
    switch(FFHIP_VP8_FRAMES_MIN=2)
    assert L.ffhip_vp8_decode_frames_form(n) == 1 and L.ffhip_vp8_decode_frames_form(1) == 0
    fused_form = ops.vp8_decode_frames(c, r, modes, resid, 2, flt)
    for i in range(n):
        exp = oracle_chain(c, r, 2, modes[i], resid[i], flt)[0]
        assert np.array_equal(rows_form[i], exp) and np.array_equal(fused_form[i], exp), i


@pytest.mark.parametrize("slack", [1, 3, 200])
def test_vp8_slack(slack, switch):
    """FFHIP_VP8_SLACK: a row that has to block on the row above waits for `slack` macroblocks more than it needs (never more than the row has)"""
    from test_vp8_lf_gpu import oracle_lf
    c, r, n, modes, resid, flt = _vp8_case(seed=1600)
    switch(FFHIP_VP8_SLACK=slack)
    got = ops.vp8_predict_loopfilter(c, r, modes, resid, 2, flt)
    for i in range(n):
        exp = oracle_lf(c, r, 2, modes[i], flt, O.oracle_vp8_frame(c, r, modes[i], resid[i]))
        for gp, e, name in zip(got, exp, "YUV"):
            assert np.array_equal(gp[i], e), (slack, i, name)


def test_diagnostics_prints_change_nothing(switch, capfd):
    """FFHIP_VERBOSE, FFHIP_PLAN_TIMES, FFHIP_HUFF_TIMES: prints on stderr, the same bytes"""
    data = open(os.path.join(os.path.dirname(__file__), "golden", FILES["q85_420"]), "rb").read()
    tus, res = synth.hevc_intra_tus(256, 192, seed=17)
    want_px = ops.jpeg_decode_files_device([data] * 3, n_threads=2)[1]
    want_y = ops.hevc_intra_recon(tus, res, 256, 192, True, 8, 8)
    switch(FFHIP_VERBOSE=1, FFHIP_PLAN_TIMES=1, FFHIP_HUFF_TIMES=1, FFHIP_JPEG_GPU_ENTROPY=1)
    got_px = ops.jpeg_decode_files_device([data] * 3, n_threads=2)[1]
    got_y = ops.hevc_intra_recon(tus, res, 256, 192, True, 8, 8)
    switch(FFHIP_HEVC_PLAN="host")
    got_yh = ops.hevc_intra_recon(tus, res, 256, 192, True, 8, 8)
    err = capfd.readouterr().err
    assert np.array_equal(got_px, want_px)
    for a, b, c in zip(got_y, want_y, got_yh):
        assert np.array_equal(a, b) and np.array_equal(c, b)
    assert "intra_recon host:" in err and "plan:" in err


@pytest.mark.parametrize("ring", [16, 32])
def test_huffman_ring_sizes(ring, switch):
    """FFHIP_HUFF_RING (with FFHIP_JPEG_SYNC=0, the lane-per-interval kernel): the device Huffman kernel's byte ring per lane (64 bytes and a refill every 8 symbols for batches beyond what two workgroups per CU
    hold, 128 bytes and every 16 otherwise): the same pixels from files with restart markers, with several tables, and from plain files forced onto the device"""
    here = os.path.join(os.path.dirname(__file__), "golden")
    for tag in ("q85_420_dri", "q85_411", "q92_444"):
        data = open(os.path.join(here, FILES[tag]), "rb").read()
        switch(FFHIP_JPEG_GPU_ENTROPY=0)
        want = ops.jpeg_decode_files_device([data] * 5, n_threads=2)[1]
        switch(FFHIP_JPEG_GPU_ENTROPY=1, FFHIP_HUFF_RING=ring, FFHIP_JPEG_SYNC=0)
        got = ops.jpeg_decode_files_device([data] * 5, n_threads=2)[1]
        assert np.array_equal(got, want), (tag, ring)


def test_plain_files_one_lane_each_and_few_rounds(switch):
    """FFHIP_JPEG_SYNC=0: a file without restart markers is one lane's (round 4's form, for batches of thousands); FFHIP_JPEG_SYNC_ROUNDS: the
    list rounds are launched one (two) at a time, the host looks whether they reached their fixed point and launches more -- the path a scan takes that does
    not settle within the ten rounds launched at once; FFHIP_JPEG_SYNC_BITS: the length of a subsequence, 128 bits (a few symbols, hardly ever in step
    after one) to the whole scan in one; FFHIP_JPEG_SYNC_PARTS: the parts a batch is staged, sent and decoded in (their kernels alternate between the caller's stream and one of
    the library's; FFHIP_JPEG_SYNC_STREAMS=1: all on the caller's)"""
    from test_huff_gpu import _plain_file, same_planes
    files = [_plain_file((360, 640), 85, seed=i) for i in range(3)] + [_plain_file((360, 640), 60, seed=i) for i in range(6)]
    want = ops.jpeg_entropy_batch_gpu(files)
    for env in ({"FFHIP_JPEG_SYNC": 0}, {"FFHIP_JPEG_SYNC_ROUNDS": 1}, {"FFHIP_JPEG_SYNC_ROUNDS": 2}, {"FFHIP_JPEG_SYNC_ROUNDS": 32}, {"FFHIP_JPEG_SYNC_BITS": 128},
                {"FFHIP_JPEG_SYNC_BITS": 128, "FFHIP_JPEG_SYNC_ROUNDS": 1}, {"FFHIP_JPEG_SYNC_BITS": 4096}, {"FFHIP_JPEG_SYNC_BITS": 65536}, {"FFHIP_JPEG_SYNC_PARTS": 1},
                {"FFHIP_JPEG_SYNC_PARTS": 3}, {"FFHIP_JPEG_SYNC_PARTS": 8}, {"FFHIP_JPEG_SYNC_PARTS": 4, "FFHIP_JPEG_SYNC_STREAMS": 1}):
        switch(**env)
        got = ops.jpeg_entropy_batch_gpu(files)
        for a, b in zip(got[1:], want[1:]):
            assert np.array_equal(a, b), env
        same_planes(files)
        switch(FFHIP_JPEG_SYNC=1, FFHIP_JPEG_SYNC_ROUNDS=10, FFHIP_JPEG_SYNC_BITS=2048, FFHIP_JPEG_SYNC_PARTS=0, FFHIP_JPEG_SYNC_STREAMS=0)
