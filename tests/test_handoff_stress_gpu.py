"""GPU: the in-launch hand-offs (VP8 rows, VP8 loop-filter rows, HEVC groups) under UNEVEN load -- a second stream
keeps the memory system busy with large copies while the dependency-scheduled kernels run, and every byte is still
the oracle's.  (MI355X_MICROARCH.md: idle chips and uniform load hide hand-off bugs.)"""
import threading

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth

pytestmark = pytest.mark.gpu


class Hog:
    """copies 256 MB back and forth on its own stream until stopped"""

    def __init__(self):
        self.L = capi.require_device()
        self.st = self.L.ffhip_stream_create()
        self.a = ops.DeviceBuffer(nbytes=256 << 20)
        self.b = ops.DeviceBuffer(nbytes=256 << 20)
        self.stop = False
        self.count = 0
        self.t = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop:
            for _ in range(8):
                capi.check(self.L.ffhip_copy_calibrate(self.b.ptr, self.a.ptr, 256 << 20, self.st))
                capi.check(self.L.ffhip_copy_calibrate(self.a.ptr, self.b.ptr, 256 << 20, self.st))
            capi.check(self.L.ffhip_stream_sync(self.st))
            self.count += 16

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        self.t.join()
        self.L.ffhip_stream_destroy(self.st)


def test_vp8_rows_and_loopfilter_under_load():
    c, r, n = 120, 68, 3
    modes = np.stack([synth.vp8_modes(c, r, seed=900 + i) for i in range(n)])
    modes[..., 18] = np.random.default_rng(9).integers(0, 4, size=modes[..., 18].shape)
    resid = np.stack([synth.vp8_residual(c * r, seed=900 + i, amplitude=60) for i in range(n)])
    flt = synth.vp8_filters(seed=3)
    exp = [O.oracle_vp8_frame(c, r, modes[i], resid[i]) for i in range(n)]
    with Hog() as hog:
        for _ in range(3):
            got = ops.vp8_predict_recon(c, r, modes, resid)
            for i in range(n):
                for gp, e, name in zip(got, exp[i], "YUV"):
                    assert np.array_equal(gp[i], e), (i, name)
            lf = ops.vp8_loopfilter(c, r, 2, modes, flt, got[0], got[1], got[2])
            for i in range(n):
                p = [np.ascontiguousarray(x).copy() for x in exp[i]]
                O.ffo().ffo_vp8_loopfilter_frame(c, r, 2, np.ascontiguousarray(modes[i]).reshape(-1), np.ascontiguousarray(flt).reshape(-1),
                                                 p[0].reshape(-1), p[1].reshape(-1), p[2].reshape(-1))
                for gp, e, name in zip(lf, p, "YUV"):
                    assert np.array_equal(gp[i], e), ("lf", i, name)
    assert hog.count > 0


def test_hevc_groups_under_load():
    w, h = 512, 384
    tus, res = synth.hevc_intra_tus(w, h, 77, adversarial_masks=True)
    exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8)
    with Hog() as hog:
        for _ in range(4):
            got = ops.hevc_intra_recon(tus, res, w, h, True, 8, 8)
            for gp, e, name in zip(got, exp, "YUV"):
                assert np.array_equal(gp, e), name
    assert hog.count > 0


def test_give_up_path_reports_eio_and_the_library_recovers(monkeypatch):
    """The bounded spin of the dependency-scheduled kernels: with the done flag of one TU withheld (test hook
    FFHIP_DEBUG_WITHHOLD_TU), the groups that wait for it run into SPIN_LIMIT, the launch drains (every wave sees the
    abort word), ffhip_stream_sync returns FFHIP_EIO exactly once, and after ffhip_shutdown the same list decodes to the
    oracle's picture again."""
    import oracle_lib as O
    from ffpic_amd import capi, ops, synth
    L = capi.require_device()
    w, h = 256, 192
    tus, res = synth.hevc_intra_tus(w, h, seed=91)
    # a TU of the first coding tree block that later groups read: the last luma TU of the first 64x64 window (the default scheduling window)
    first = [i for i, t in enumerate(tus) if t["cidx"] == 0 and t["x"] < 64 and t["y"] < 64]
    victim = first[-1]
    monkeypatch.setenv("FFHIP_DEBUG_WITHHOLD_TU", str(victim)); capi.reload_env()
    with pytest.raises(capi.FfhipError) as ei:
        ops.hevc_intra_recon(tus, res, w, h, True, 8, 8)           # the wrapper's stream sync sees the abort
    assert "-5" in str(ei.value)
    assert L.ffhip_stream_sync(None) == 0                           # reported once, then clear
    monkeypatch.delenv("FFHIP_DEBUG_WITHHOLD_TU"); capi.reload_env()
    L.ffhip_shutdown()
    capi.require_device()
    got = ops.hevc_intra_recon(tus, res, w, h, True, 8, 8)
    exp = O.oracle_hevc_intra(tus, res, w, h, True, 8, 8)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)


def test_side_by_side_vp8_call_is_repeated_by_the_sync(monkeypatch):
    """ffhip_vp8_predict_loopfilter on a device where the filter's wait for the prediction runs out (test hook FFHIP_DEBUG_VP8_LF_GIVEUP: the fused
    filter launch gives up half-way down every frame): ffhip_stream_sync puts the one column of the planes' former contents the prediction
    reads back, runs prediction and filter one after the other and returns FFHIP_RETRIED (> 0: outputs good, consumers behind the call stale)
    with the bytes of an undisturbed call -- H_PRED in the first column included, whose wrapped read sees that column.  The abort is reported in
    a word of the call's own: the process-wide word stays clear, and a later abort of ANOTHER kernel does not set the retry off again.  With
    FFHIP_VP8_NO_RETRY the sync says FFHIP_EIO, as before round 4."""
    import oracle_lib as O
    from ffpic_amd import capi, ops, synth
    from test_vp8_lf_gpu import oracle_lf
    L = capi.require_device()
    c, r, n = 21, 13, 3
    modes = np.stack([synth.vp8_modes(c, r, seed=1200 + i) for i in range(n)])
    modes.reshape(n, r, c, 20)[:, 1::2, 0, 0] = 3              # every other row starts with the wrapped H_PRED
    resid = np.stack([synth.vp8_residual(c * r, seed=1210 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=31)
    exp = []
    for i in range(n):
        y0, u0, v0 = O.oracle_vp8_frame(c, r, modes[i], resid[i])
        exp.append(oracle_lf(c, r, 2, modes[i], flt, (y0, u0, v0)))
    want = ops.vp8_predict_loopfilter(c, r, modes, resid, 2, flt)          # undisturbed
    assert ops.last_sync_status == 0
    monkeypatch.setenv("FFHIP_DEBUG_VP8_LF_GIVEUP", "1"); capi.reload_env()
    got = ops.vp8_predict_loopfilter(c, r, modes, resid, 2, flt)           # gives up, is repeated inside the wrapper's ffhip_stream_sync
    assert ops.last_sync_status == capi.FFHIP_RETRIED
    for i in range(n):
        for gp, wp, e, name in zip(got, want, exp[i], "YUV"):
            assert np.array_equal(gp[i], e) and np.array_equal(wp[i], e), (i, name)
    assert L.ffhip_stream_sync(None) == 0
    # an abort of some OTHER dependency-scheduled kernel on the stream afterwards is that kernel's FFHIP_EIO, not a second repeat of this call
    tus, res = synth.hevc_intra_tus(256, 192, seed=91)
    first = [i for i, t in enumerate(tus) if t["cidx"] == 0 and t["x"] < 64 and t["y"] < 64]
    monkeypatch.setenv("FFHIP_DEBUG_WITHHOLD_TU", str(first[-1])); capi.reload_env()
    with pytest.raises(capi.FfhipError) as ei:
        ops.hevc_intra_recon(tus, res, 256, 192, True, 8, 8)
    assert "-5" in str(ei.value)
    monkeypatch.delenv("FFHIP_DEBUG_WITHHOLD_TU"); capi.reload_env()
    assert L.ffhip_stream_sync(None) == 0
    monkeypatch.setenv("FFHIP_VP8_NO_RETRY", "1"); capi.reload_env()
    with pytest.raises(capi.FfhipError) as ei:
        ops.vp8_predict_loopfilter(c, r, modes, resid, 2, flt)
    assert "-5" in str(ei.value)
    assert L.ffhip_stream_sync(None) == 0
    monkeypatch.delenv("FFHIP_VP8_NO_RETRY"); monkeypatch.delenv("FFHIP_DEBUG_VP8_LF_GIVEUP"); capi.reload_env()
    again = ops.vp8_predict_loopfilter(c, r, modes, resid, 2, flt)
    assert ops.last_sync_status == 0
    for i in range(n):
        for gp, e in zip(again, exp[i]):
            assert np.array_equal(gp[i], e)


@pytest.mark.parametrize("planes", [False, True])
def test_decode_frames_row_form_repeats_its_colour_conversion(monkeypatch, planes):
    """ffhip_vp8_decode_frames in its row form (prediction || filter, then the planar colour kernel on the same stream) with the filter giving up
    (FFHIP_DEBUG_VP8_LF_GIVEUP): the colour conversion has converted the ABORTED planes by the time ffhip_stream_sync repeats prediction and
    filter -- it belongs to the call, so the retry runs it again behind the filter, and the BGRA is the oracle's (round 4 returned FFHIP_OK with
    the broken run's pixels).  The status is FFHIP_RETRIED all the same: the caller may have enqueued consumers of the BGRA behind the call."""
    from ffpic_amd import capi, ops, synth
    from test_vp8_frames_gpu import oracle_chain
    capi.require_device()
    c, r, n = 21, 13, 3
    modes = np.stack([synth.vp8_modes(c, r, seed=1300 + i) for i in range(n)])
    modes.reshape(n, r, c, 20)[:, 1::2, 0, 0] = 3
    resid = np.stack([synth.vp8_residual(c * r, seed=1310 + i) for i in range(n)])
    flt = synth.vp8_filters(seed=32)
    exp = [oracle_chain(c, r, 2, modes[i], resid[i], flt)[0] for i in range(n)]
    monkeypatch.setenv("FFHIP_VP8_FRAMES", "rows"); capi.reload_env()
    out = ops.vp8_decode_frames(c, r, modes, resid, 2, flt, planes=planes)
    assert ops.last_sync_status == 0
    bgra = out[0] if planes else out
    for i in range(n):
        assert np.array_equal(bgra[i], exp[i]), i
    monkeypatch.setenv("FFHIP_DEBUG_VP8_LF_GIVEUP", "1"); capi.reload_env()
    out = ops.vp8_decode_frames(c, r, modes, resid, 2, flt, planes=planes)
    assert ops.last_sync_status == capi.FFHIP_RETRIED
    bgra = out[0] if planes else out
    for i in range(n):
        assert np.array_equal(bgra[i], exp[i]), i
    monkeypatch.delenv("FFHIP_DEBUG_VP8_LF_GIVEUP"); monkeypatch.delenv("FFHIP_VP8_FRAMES"); capi.reload_env()
