"""GPU: the JPEG entropy front end on the device (both forms, see `form` below) against the host decoder,
whose planes the CPU tests pin to the reference's whole-file decode -- and (round 4, bottom of the file) against the
reference's whole-file decode directly."""
import io
import os

import numpy as np
import pytest

from ffpic_amd import capi, ops

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(params=["subsequences", "lane_per_interval"], autouse=True)
def form(request, monkeypatch):
    """every test of this file with both device decoders: the subsequence decoder (round 5, the default: a lane per 2048 bits of a restart interval,
    synchronised over rounds) and the kernel of rounds 3-4 (FFHIP_JPEG_SYNC=0: a lane per restart interval, a file without markers one lane's)"""
    if request.param == "lane_per_interval":
        monkeypatch.setenv("FFHIP_JPEG_SYNC", "0")
    else:
        monkeypatch.setenv("FFHIP_JPEG_SYNC", "1")      # (unset, batches whose restart intervals are a subsequence or two long take the other kernel)
    capi.reload_env()
    yield request.param
    monkeypatch.undo()
    capi.reload_env()


def same_planes(files):
    g, cy, cu, cv, q = ops.jpeg_entropy_batch_gpu(files)
    g2, hy, hu, hv, hq = ops.jpeg_entropy_batch(files, n_threads=2)
    assert np.array_equal(q, hq)
    assert np.array_equal(cy, hy)
    if g.ncomp == 3:
        assert np.array_equal(cu, hu) and np.array_equal(cv, hv)
    return g


def test_dri_fixture():
    data = open(os.path.join(GOLDEN, "file_q85_420_dri.jpg"), "rb").read()
    same_planes([data])
    same_planes([data] * 5)


def test_files_without_restart_markers():
    """no DRI: the scan is cut into subsequences of 2048 bits, a lane each, that synchronise with each other (round 5)"""
    for name in ("file_q85_420.jpg", "file_q92_444.jpg", "file_q80_grey.jpg", "file_q85_411.jpg", "file_q85_114.jpg", "file_q88_422.jpg"):
        data = open(os.path.join(GOLDEN, name), "rb").read()
        same_planes([data])
        same_planes([data] * 3)


def _plain_file(size, q, sub=2, mode="RGB", seed=0, noise=25.0, optimize=False):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size[0], 0:size[1]]
    img = np.stack([128 + 100 * np.sin(xx / (9.0 + seed)), 128 + 90 * np.cos(yy / 7.0), (xx * 3 + yy * 5) % 256], axis=2)
    img = np.clip(img + rng.normal(0, noise, img.shape), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    kw = dict(quality=q, optimize=optimize)
    if mode == "RGB":
        kw["subsampling"] = sub
    PIL.fromarray(img).convert(mode).save(bio, "JPEG", **kw)
    data = bio.getvalue()
    assert b"\xff\xdd" not in data
    return data


@pytest.mark.parametrize("size,q,sub,mode,noise,optimize", [
    ((200, 296), 90, 2, "RGB", 25.0, False),      # a few dozen subsequences
    ((1080, 1920), 85, 2, "RGB", 25.0, False),    # thousands, several workgroups per picture
    ((1080, 1920), 100, 0, "RGB", 40.0, False),   # long codes and magnitudes: few symbols per subsequence
    ((1080, 1920), 20, 2, "RGB", 0.0, False),     # smooth at low quality: whole rows of blocks inside one subsequence
    ((600, 808), 75, 1, "RGB", 25.0, True),       # 4:2:2, the file's own optimised tables (a set per file)
    ((600, 808), 85, 0, "L", 25.0, True),         # one component
    ((8, 8), 90, 2, "RGB", 25.0, False),          # one MCU: less than one subsequence
])
def test_generated_files_without_restart_markers(size, q, sub, mode, noise, optimize):
    files = [_plain_file(size, q, sub, mode, seed=i, noise=noise, optimize=optimize) for i in range(3)]
    same_planes(files)
    same_planes(files[1:2])


def _pixels(files, **env):
    """files -> BGRA through ffhip_jpeg_decode_files_device with library switches set for the call"""
    import os as _os
    old = {k: _os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            _os.environ[k] = str(v)
        capi.reload_env()
        return ops.jpeg_decode_files_device(files, n_threads=4)[1]
    finally:
        for k, v in old.items():
            if v is None:
                _os.environ.pop(k, None)
            else:
                _os.environ[k] = v
        capi.reload_env()


@pytest.mark.parametrize("n", [1, 5, 40, 130, 300])
def test_plain_files_to_pixels_in_parts(n):
    """files without restart markers -> BGRA on the device: the batch goes in parts (of about 140 MB of scan bytes by default; forced here), every
    part's reconstruction enqueued by the entropy call behind the part's write pass.  Same pixels as with the entropy decode on the host."""
    files = [_plain_file((64, 96), 70 + (i % 4) * 7, seed=i % 9, optimize=bool(i % 3 == 0)) for i in range(n)]
    want = _pixels(files, FFHIP_JPEG_GPU_ENTROPY=0)
    assert np.array_equal(_pixels(files), want)                                   # (a few megabytes: one part by default)
    for parts in (2, 3, 5):
        assert np.array_equal(_pixels(files, FFHIP_JPEG_SYNC_PARTS=parts), want), parts


def test_batch_of_files_with_and_without_restart_markers():
    """one file without DRI among files with: the batch takes the lane-per-interval kernel, the plain file one lane"""
    files = [_good_dri_file(blocks=3), _good_dri_file(blocks=0), _good_dri_file(blocks=5)]
    assert b"\xff\xdd" not in files[1]
    same_planes(files)
    assert np.array_equal(_pixels(files), _pixels(files, FFHIP_JPEG_GPU_ENTROPY=0))


def test_damaged_plain_file_in_a_batch_goes_to_the_host_decoder():
    """a plain file with bytes of its scan replaced: the device decoder refuses the batch, ffhip_jpeg_decode_files_device falls back to the host decoder
    for it -- the same pixels and the same per-picture verdicts as with FFHIP_JPEG_GPU_ENTROPY=0, whatever those are"""
    good = _plain_file((128, 160), 85, seed=3)
    bad = bytearray(good)
    sos = good.find(b"\xff\xda")
    rng = np.random.default_rng(9)
    for k in rng.integers(sos + 40, len(bad) - 8, size=12):
        if bad[k] != 0xFF and bad[k - 1] != 0xFF:
            bad[k] ^= 0x55
    files = [good, bytes(bad), good]
    try:
        want = _pixels(files, FFHIP_JPEG_GPU_ENTROPY=0)
    except capi.FfhipError:
        with pytest.raises(capi.FfhipError):
            _pixels(files)
    else:
        assert np.array_equal(_pixels(files), want)
    assert np.array_equal(_pixels([good] * 2), _pixels([good] * 2, FFHIP_JPEG_GPU_ENTROPY=0))


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_random_corruption_of_plain_files_never_faults_and_is_never_silently_different(seed):
    """seeded byte corruptions inside the scan of files WITHOUT restart markers, several subsequences long: every call returns.  The subsequence decoder
    reads garbage by design (every lane starts from a guess), so what matters is the end: a batch it accepts has the host decoder's planes, bit for bit,
    and one the host decoder refuses is refused"""
    rng = np.random.default_rng(seed)
    good = _plain_file((240, 320), 88, seed=seed % 5, noise=35.0)
    sos = good.find(b"\xff\xda")
    start = sos + 2 + ((good[sos + 2] << 8) | good[sos + 3])
    for _ in range(6):
        files = []
        for _ in range(5):
            d = bytearray(good)
            for _ in range(int(rng.integers(0, 6))):
                k = int(rng.integers(start, len(d) - 2))
                if d[k] != 0xFF and d[k - 1] != 0xFF:
                    d[k] = int(rng.integers(0, 255))       # never creates or destroys a 0xFF
            files.append(bytes(d))
        try:
            g, cy, cu, cv, q = ops.jpeg_entropy_batch_gpu(files)
        except capi.FfhipError:
            continue
        try:
            g2, hy, hu, hv, hq = ops.jpeg_entropy_batch(files, n_threads=2)
        except capi.FfhipError:
            # the host decoder's truncation rule looks at the bytes fed behind the data, the device's at the bit position of the last block: a damaged
            # file that ends within its last byte may pass one and not the other.  Nothing to compare then.
            continue
        assert np.array_equal(cy, hy) and np.array_equal(cu, hu) and np.array_equal(cv, hv)
    same_planes([good] * 2)


def test_many_small_plain_files_share_workgroups():
    """pictures of a few subsequences each: a workgroup's 256 lanes span many pictures, with the tables of the first in LDS"""
    files = [_plain_file((48, 64), 60 + (i % 5) * 8, seed=i, optimize=bool(i & 1)) for i in range(40)]
    same_planes(files)


def test_other_geometry_is_refused():
    a = open(os.path.join(GOLDEN, "file_q85_420.jpg"), "rb").read()
    b = open(os.path.join(GOLDEN, "file_q85_420_dri.jpg"), "rb").read()
    with pytest.raises(capi.FfhipError):
        ops.jpeg_entropy_batch_gpu([a, b])


@pytest.mark.parametrize("sub,mode,blocks,q", [(2, "RGB", 1, 90), (2, "RGB", 7, 60), (0, "RGB", 3, 95), (1, "RGB", 5, 75), (0, "L", 4, 85),
                                                (2, "RGB", 40, 30), (2, "RGB", 2, 100)])
def test_generated_files(sub, mode, blocks, q):
    """PIL-made files: 4:2:0 / 4:4:4 / 4:2:2 / grey, short and long restart intervals, low and high quality
    (quality 100 exercises long codes and 16-bit-ish magnitudes, quality 30 long zero runs)"""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(blocks * 100 + q)
    files = []
    for i in range(3):
        yy, xx = np.mgrid[0:200, 0:296]
        img = np.stack([128 + 100 * np.sin(xx / (9.0 + i)), 128 + 90 * np.cos(yy / 7.0), (xx * 3 + yy * 5) % 256], axis=2)
        img = np.clip(img + rng.normal(0, 25, img.shape), 0, 255).astype(np.uint8)
        bio = io.BytesIO()
        im = PIL.fromarray(img).convert(mode)
        kw = dict(quality=q, restart_marker_blocks=blocks)
        if mode == "RGB":
            kw["subsampling"] = sub
        im.save(bio, "JPEG", **kw)
        files.append(bio.getvalue())
    assert b"\xff\xdd" in files[0]
    same_planes(files)


def test_corrupt_interval_is_reported_not_crashed():
    """bytes of one interval replaced by noise: that picture is flagged, the call returns an error, nothing faults"""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    img = np.clip(rng.normal(128, 50, (128, 160, 3)), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    PIL.fromarray(img).save(bio, "JPEG", quality=80, subsampling=2, restart_marker_blocks=2)
    data = bytearray(bio.getvalue())
    k = bytes(data).find(b"\xff\xd2")
    assert k > 0
    noise = rng.integers(0, 255, 40).astype(np.uint8)      # no 0xFF: the marker structure survives
    data[k + 2:k + 42] = bytes(noise)
    try:
        ops.jpeg_entropy_batch_gpu([bytes(data)])
    except capi.FfhipError:
        pass                                               # flagged: fine; decoding garbage without a flag is fine too


def test_random_corruption_never_faults():
    """seeded byte corruptions inside the entropy-coded data (markers left alone): every call returns, with or
    without an error; the device decoder's reads and writes are all bounded by the lane's interval and block"""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    img = np.clip(rng.normal(128, 60, (96, 128, 3)), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    PIL.fromarray(img).save(bio, "JPEG", quality=90, subsampling=2, restart_marker_blocks=3)
    good = bio.getvalue()
    sos = good.find(b"\xff\xda")
    start = sos + 2 + ((good[sos + 2] << 8) | good[sos + 3])
    files = []
    for _ in range(24):
        d = bytearray(good)
        for _ in range(int(rng.integers(1, 12))):
            k = int(rng.integers(start, len(d) - 2))
            if d[k] != 0xFF and d[k - 1] != 0xFF:
                d[k] = int(rng.integers(0, 255))       # never creates or destroys a 0xFF
        files.append(bytes(d))
    try:
        ops.jpeg_entropy_batch_gpu(files)
    except capi.FfhipError:
        pass
    same_planes([good])                                 # and the device is still fine afterwards


def _good_dri_file(blocks=3, size=(96, 128)):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    img = np.clip(rng.normal(128, 60, size + (3,)), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    kw = dict(quality=90, subsampling=2)
    if blocks:
        kw["restart_marker_blocks"] = blocks
    PIL.fromarray(img).save(bio, "JPEG", **kw)
    return bio.getvalue()


def test_truncated_file_without_restart_markers_is_refused():
    """One lane owns the whole scan of a file without DRI.  Cut short, the lane must stop at the end of ITS bytes
    (zeros behind them, then a flag), not walk on through the next picture's data, the tables and the end of the
    device scratch."""
    good = _good_dri_file(blocks=0, size=(256, 320))
    assert b"\xff\xdd" not in good
    for keep in (0.5, 0.1):
        cut = good[: int(len(good) * keep)]
        with pytest.raises(capi.FfhipError):
            ops.jpeg_entropy_batch_gpu([cut, good, cut])
    same_planes([good] * 2)


def test_empty_restart_intervals_are_refused():
    """every RSTn present, but the intervals between two of them carry no data: the lanes of the empty intervals have
    nothing of their own to read"""
    good = _good_dri_file(blocks=3)
    d = bytearray(good)
    a, b = bytes(d).find(b"\xff\xd1"), bytes(d).find(b"\xff\xd2")
    assert 0 < a < b
    del d[a + 2:b]                              # RST1 directly followed by RST2
    with pytest.raises(capi.FfhipError):
        ops.jpeg_entropy_batch_gpu([bytes(d)])
    # all data removed: the scan is nothing but its markers
    sos = good.find(b"\xff\xda")
    start = sos + 2 + ((good[sos + 2] << 8) | good[sos + 3])
    n_rst = sum(good.count(bytes([0xFF, 0xD0 + k])) for k in range(8))
    hollow = good[:start] + b"".join(bytes([0xFF, 0xD0 + (k & 7)]) for k in range(n_rst)) + b"\xff\xd9"
    with pytest.raises(capi.FfhipError):
        ops.jpeg_entropy_batch_gpu([hollow])
    same_planes([good])


def test_hostile_tables_cannot_read_past_the_staged_bytes():
    """Complete 1-bit codes that map to DC size 11 and AC (run 0, size 15): no symbol ever ends a block early and none is
    invalid, so a 2048x2048 picture asks for ~30 MB of bits from a 40-byte scan.  With unbounded reads that decoded
    whatever followed in device memory; now the lane is flagged within a few dwords of its interval's end."""
    def seg(marker, body):
        return bytes([0xFF, marker]) + (len(body) + 2).to_bytes(2, "big") + body
    dqt = seg(0xDB, b"\x00" + bytes([1] * 64))
    sof = seg(0xC0, b"\x08" + (2048).to_bytes(2, "big") + (2048).to_bytes(2, "big") + b"\x03" + b"\x01\x22\x00\x02\x11\x00\x03\x11\x00")
    counts = bytes([2]) + bytes(15)
    dht = seg(0xC4, b"\x00" + counts + b"\x0b\x0b") + seg(0xC4, b"\x10" + counts + b"\x0f\x0f")
    sos = seg(0xDA, b"\x03\x01\x00\x02\x00\x03\x00\x00\x3f\x00")
    rng = np.random.default_rng(2)
    scan = bytes(int(x) for x in rng.integers(0, 255, 40))      # no 0xFF
    hostile = b"\xff\xd8" + dqt + sof + dht + sos + scan + b"\xff\xd9"
    g, w, h = ops.jpeg_probe(hostile)
    assert (g.mcu_cols, g.mcu_rows) == (128, 128)
    with pytest.raises(capi.FfhipError):
        ops.jpeg_entropy_batch_gpu([hostile] * 4)
    with pytest.raises(capi.FfhipError):
        ops.jpeg_entropy_batch([hostile])                       # the host decoder refuses it as well
    same_planes([_good_dri_file()])


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: the device front end against the REFERENCE, not against the host decoder.  tests/golden/jpeg_files.npz,
# jpeg_file_422.npz and jpeg_file_411.npz hold what the reference's own loader (format/jpg.c:255-415 decode_data_unit,
# :588-637 read_compressed_scan, coding/huffman.c:92-222) made of each fixture file: BGRA or its SHA-256.  Two files
# (the small ones whose last data unit the reference's bit reader runs dry in, utils/bitstream.c:117) compare
# everything except the last MCU, exactly as test_jpeg_gpu.py::test_golden_files does for the reconstruction alone.
# ---------------------------------------------------------------------------------------------------------------------
import ctypes as C
import hashlib

from test_oracle_golden import FILES


def _equals_reference_decode(g, tag, geom, img):
    H, W = [int(x) for x in g[f"{tag}_shape"][:2]]
    img = np.ascontiguousarray(img[:H, :W])
    if int(g[f"{tag}_last_mcu_exact"]):
        assert hashlib.sha256(img.tobytes()).digest() == g[f"{tag}_sha256"].tobytes(), tag
    else:
        keep = np.ones((H, W), bool)
        keep[(geom.mcu_rows - 1) * 8 * geom.v:, (geom.mcu_cols - 1) * 8 * geom.h:] = False
        assert np.array_equal(img[keep], g[f"{tag}_bgra"][keep]), tag


@pytest.mark.parametrize("tag", list(FILES))
@pytest.mark.parametrize("device_entropy", ["auto", "1", "0"])
def test_files_to_device_pixels_equal_the_reference_decode(golden, monkeypatch, tag, device_entropy):
    """ffhip_jpeg_decode_files_device on every fixture file -> BGRA in device memory -> the reference's whole-file decode.
    "1" sends files WITHOUT restart markers through the device Huffman kernel too (one lane per file); "0" keeps every
    file on the host threads; "auto" is the shipped choice (device for DRI files)."""
    if device_entropy != "auto":
        monkeypatch.setenv("FFHIP_JPEG_GPU_ENTROPY", device_entropy)
    else:
        monkeypatch.delenv("FFHIP_JPEG_GPU_ENTROPY", raising=False)
    capi.reload_env()
    g = golden("jpeg_files.npz")
    data = open(os.path.join(GOLDEN, FILES[tag]), "rb").read()
    geom, out, _ = ops.jpeg_decode_files_device([data] * 3, n_threads=2)
    for i in range(3):
        _equals_reference_decode(g, tag, geom, out[i])


@pytest.mark.parametrize("tag", list(FILES))
def test_device_entropy_planes_reconstruct_to_the_reference_decode(golden, tag):
    """ffhip_jpeg_entropy_batch_gpu leaves coefficient planes and quantisers in device memory; ffhip_jpeg_recon_batch takes
    them from there (nothing visits the host in between) -> the reference's whole-file decode."""
    L = capi.require_device()
    g = golden("jpeg_files.npz")
    data = open(os.path.join(GOLDEN, FILES[tag]), "rb").read()
    n = 2
    geom, _, _ = ops.jpeg_probe(data)
    bufs = [np.frombuffer(data, dtype=np.uint8) for _ in range(n)]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    dy = ops.DeviceBuffer(nbytes=n * geom.y_blocks * 128)
    du = ops.DeviceBuffer(nbytes=max(n * geom.c_blocks * 128, 16))
    dv = ops.DeviceBuffer(nbytes=max(n * geom.c_blocks * 128, 16))
    dq = ops.DeviceBuffer(nbytes=n * 512)
    status = (C.c_int * n)()
    three = geom.ncomp == 3
    capi.check(L.ffhip_jpeg_entropy_batch_gpu(ptrs, lens, n, 2, C.byref(geom), dy.ptr, du.ptr if three else None,
                                              dv.ptr if three else None, dq.ptr, status, None), "ffhip_jpeg_entropy_batch_gpu")
    assert list(status) == [0] * n
    H, W = geom.height, geom.width
    dout = ops.DeviceBuffer(nbytes=n * H * W * 4)
    ops.jpeg_recon_batch(geom, n, dy.ptr, du.ptr if three else None, dv.ptr if three else None, dq.ptr, 256, dout.ptr, W * 4, H * W * 4)
    capi.check(L.ffhip_stream_sync(None))
    out = dout.to_host((n, H, W, 4), np.uint8)
    for i in range(n):
        _equals_reference_decode(g, tag, geom, out[i])


def test_three_threads_each_on_its_own_stream():
    """callers on different streams overlap completely: staging, device scratch, the upload stream and its events are per stream or per thread.
    Three threads decode different batches (files with and without restart markers) several times over; every result is the single-threaded one."""
    import ctypes as C
    import threading
    torch = pytest.importorskip("torch")
    L = capi.require_device()
    batches = [[_plain_file((96, 128), 80, seed=i) for i in range(9)], [_good_dri_file(blocks=3)] * 7 + [_good_dri_file(blocks=0)] * 2,
               [_plain_file((96, 128), 92, seed=i, optimize=True) for i in range(33)]]
    want = [_pixels(b) for b in batches]
    errors = []

    def worker(k):
        try:
            files = batches[k]
            n = len(files)
            g, _, _ = ops.jpeg_probe(files[0])
            bufs = [np.frombuffer(f, dtype=np.uint8) for f in files]
            ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
            lens = (C.c_size_t * n)(*[b.size for b in bufs])
            status = (C.c_int * n)()
            g2 = capi.JpegGeom()
            st = torch.cuda.Stream()
            out = torch.empty((n, g.height, g.width, 4), dtype=torch.uint8, device="cuda:0")
            for _ in range(6):
                out.zero_()
                torch.cuda.synchronize()
                capi.check(L.ffhip_jpeg_decode_files_device(ptrs, lens, n, 2, C.byref(g2), out.data_ptr(), g.width * 4, g.width * 4 * g.height, status, st.cuda_stream))
                capi.check(L.ffhip_stream_sync(st.cuda_stream))
                if not np.array_equal(out.cpu().numpy(), want[k]):
                    errors.append((k, "pixels differ"))
        except Exception as e:      # noqa: BLE001 -- reported below, from the main thread
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_short_restart_intervals_take_the_lane_per_interval_kernel_by_default(monkeypatch):
    """a DRI of one MCU: thousands of intervals of a few dozen bytes.  As shipped (FFHIP_JPEG_SYNC unset) such a batch goes to the lane-per-interval kernel
    -- every interval is a lane that starts from the truth --; forced either way it decodes to the same planes"""
    files = [_good_dri_file(blocks=1, size=(128, 160))] * 3
    assert files[0].count(b"\xff\xd0") > 5
    monkeypatch.delenv("FFHIP_JPEG_SYNC", raising=False)
    capi.reload_env()
    want = ops.jpeg_entropy_batch_gpu(files)
    same_planes(files)
    for v in ("0", "1"):
        monkeypatch.setenv("FFHIP_JPEG_SYNC", v)
        capi.reload_env()
        got = ops.jpeg_entropy_batch_gpu(files)
        for a, b in zip(got[1:], want[1:]):
            assert np.array_equal(a, b), v


@pytest.mark.parametrize("seed", range(40))
def test_random_geometries_qualities_and_restart_intervals(seed):
    """a soak over what an encoder can be asked for: random sizes (down to less than an MCU), qualities 3..100, 4:4:4 / 4:2:2 / 4:2:0 / grey, the encoder's
    or optimised tables, no restart markers / every block / every few blocks / every MCU row -- batches of four files of one geometry with different content,
    the device decoder's planes against the host decoder's"""
    PIL = pytest.importorskip("PIL.Image")
    from PIL import ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 22)      # (the encoder's output buffer: optimised tables on a noisy picture overrun the default)
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(1, 420)), int(rng.integers(1, 640))
    mode = "L" if rng.random() < 0.2 else "RGB"
    kw = dict(quality=int(rng.choice([3, 15, 40, 75, 90, 97, 100])), optimize=bool(rng.random() < 0.4))
    if mode == "RGB":
        kw["subsampling"] = int(rng.integers(0, 3))
    r = rng.random()
    if r < 0.25:
        kw["restart_marker_blocks"] = int(rng.integers(1, 12))
    elif r < 0.4:
        kw["restart_marker_rows"] = int(rng.integers(1, 4))
    files = []
    for i in range(4):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(xx / (3.0 + 5 * i) + seed), 128 + 90 * np.cos(yy / (2.0 + 3 * i)), (xx * (3 + i) + yy * 5) % 256], axis=2)
        img = np.clip(img + rng.normal(0, float(rng.choice([0, 5, 30, 80])), img.shape), 0, 255).astype(np.uint8)
        bio = io.BytesIO()
        PIL.fromarray(img).convert(mode).save(bio, "JPEG", **kw)
        files.append(bio.getvalue())
    same_planes(files)
