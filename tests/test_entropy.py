"""CPU: the C host-side JPEG front end and the BMP sink (f2).

The entropy decoder must produce exactly the coefficient planes of the fixture decoder
(tests/jpeg_entropy.py), whose planes in turn reproduce the reference's whole-file decode
(tests/test_oracle_golden.py::test_jpeg_files_config1); the BMP writer must produce the
bytes of the reference's display/bmpwriter.c."""
import ctypes as C
import io
import os

import numpy as np
import pytest

import jpeg_entropy
import oracle_lib as O
from conftest import GOLDEN
from ffpic_amd import capi, ops
from test_oracle_golden import FILES


@pytest.mark.parametrize("tag", list(FILES))
def test_fixture_files_match_reference_planes(tag):
    data = open(os.path.join(GOLDEN, FILES[tag]), "rb").read()
    dec = jpeg_entropy.decode(data)
    g, cy, cu, cv, quant = ops.jpeg_entropy_batch([data])
    assert (g.mcu_cols, g.mcu_rows, g.ncomp, g.h, g.v) == (dec["mcu_cols"], dec["mcu_rows"], dec["ncomp"], dec["h"], dec["v"])
    assert tuple(g.qt_id)[:dec["ncomp"]] == tuple(dec["qt_id"])[:dec["ncomp"]]
    assert np.array_equal(quant[0], dec["quant"])
    assert np.array_equal(cy, dec["coef"][0])
    if dec["ncomp"] == 3:
        assert np.array_equal(cu, dec["coef"][1]) and np.array_equal(cv, dec["coef"][2])


def test_restart_intervals_16bit_dqt_and_batch_threads():
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    files = []
    for i in range(7):
        img = np.clip(rng.normal(128, 50, size=(96, 128, 3)), 0, 255).astype(np.uint8)
        bio = io.BytesIO()
        PIL.fromarray(img).save(bio, "JPEG", quality=60 + 5 * i, subsampling=2, restart_marker_blocks=5)
        files.append(bio.getvalue())
    assert b"\xff\xdd" in files[0]          # DRI present
    g, cy, cu, cv, quant = ops.jpeg_entropy_batch(files, n_threads=3)
    per = g.y_blocks * 64
    for i, f in enumerate(files):
        dec = jpeg_entropy.decode(f)
        assert np.array_equal(cy[i * per:(i + 1) * per], dec["coef"][0]), i
        assert np.array_equal(cu[i * per // 4:(i + 1) * per // 4], dec["coef"][1]), i
        assert np.array_equal(quant[i], dec["quant"]), i
    # one thread and many threads agree
    g2, cy2, _, _, _ = ops.jpeg_entropy_batch(files, n_threads=16)
    assert np.array_equal(cy, cy2)


def test_rejects_what_it_does_not_decode():
    PIL = pytest.importorskip("PIL.Image")
    L = capi.lib()
    g = capi.JpegGeom()
    for bad in (b"", b"\xff\xd8", b"not a jpeg at all", b"\xff\xd8\xff\xd9"):
        buf = np.frombuffer(bad + b"\0" * 8, dtype=np.uint8)
        assert L.ffhip_jpeg_probe(buf.ctypes.data, len(bad), C.byref(g), None, None) == -22
    bio = io.BytesIO()
    PIL.fromarray(np.zeros((32, 32, 3), np.uint8)).save(bio, "JPEG", progressive=True)
    buf = np.frombuffer(bio.getvalue(), dtype=np.uint8)
    assert L.ffhip_jpeg_probe(buf.ctypes.data, buf.size, C.byref(g), None, None) == -22   # SOF2
    # truncated scan: an error, never a crash
    data = open(os.path.join(GOLDEN, FILES["q85_420"]), "rb").read()
    g, w, h = ops.jpeg_probe(data)
    cy = np.zeros(g.y_blocks * 64, np.int16); cu = np.zeros(g.c_blocks * 64, np.int16); cv = cu.copy()
    q = np.zeros((4, 64), np.uint16)
    cut = np.frombuffer(data[: len(data) // 2], dtype=np.uint8)
    rc = L.ffhip_jpeg_entropy_decode(cut.ctypes.data, cut.size, None, cy.ctypes.data, cu.ctypes.data, cv.ctypes.data, q.ctypes.data)
    assert rc == -22                        # a stream that runs dry is refused (no zeros decoded as data)
    # geometry mismatch against the batch geometry
    other = capi.jpeg_geom(g.mcu_cols + 1, g.mcu_rows)
    full = np.frombuffer(data, dtype=np.uint8)
    assert L.ffhip_jpeg_entropy_decode(full.ctypes.data, full.size, C.byref(other), cy.ctypes.data, cu.ctypes.data,
                                       cv.ctypes.data, q.ctypes.data) == -22


def test_oversubscribed_dht_and_zero_dimensions_are_refused():
    """A DHT whose code counts violate Kraft's inequality would put canonical codes outside their length (and outside
    the 9-bit look-up table); a SOF with width or height 0 has no MCUs.  Both are refused at parse time."""
    L = capi.lib()
    g = capi.JpegGeom()
    for counts0 in (3, 200, 255):
        counts = bytes([counts0]) + bytes(15)
        dht = b"\xff\xc4" + (2 + 1 + 16 + counts0).to_bytes(2, "big") + b"\x00" + counts + bytes(range(counts0 % 256))[:counts0].ljust(counts0, b"\x01")
        f = np.frombuffer(b"\xff\xd8" + dht + b"\xff\xd9" + bytes(16), np.uint8)
        assert L.ffhip_jpeg_probe(f.ctypes.data, f.size - 16, C.byref(g), None, None) == -22
    data = bytearray(open(os.path.join(GOLDEN, FILES["q85_420"]), "rb").read())
    k = data.find(b"\xff\xc0")
    assert k > 0
    for off in (5, 7):                      # height, width
        bad = bytearray(data)
        bad[k + off] = bad[k + off + 1] = 0
        f = np.frombuffer(bytes(bad), np.uint8)
        assert L.ffhip_jpeg_probe(f.ctypes.data, f.size, C.byref(g), None, None) == -22
    # the device front end refuses a degenerate geometry before it divides by its MCU count (no GPU needed to get there)
    zero = capi.jpeg_geom(0, 4)
    f = np.frombuffer(bytes(data), np.uint8)
    files = (C.c_void_p * 1)(f.ctypes.data)
    lens = (C.c_size_t * 1)(f.size)
    st = (C.c_int * 1)()
    assert L.ffhip_jpeg_entropy_batch_gpu(files, lens, 1, 1, C.byref(zero), 8, 8, 8, 8, st, None) == -22


def test_failed_picture_does_not_keep_stale_planes():
    """A rejected file still occupies its place in the batch: its planes come back zeroed, not with what the buffers held."""
    PIL = pytest.importorskip("PIL.Image")
    good = open(os.path.join(GOLDEN, FILES["q85_420"]), "rb").read()
    g, _, _ = ops.jpeg_probe(good)
    files = [good, good[: len(good) // 2], good]
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    arr = (C.c_void_p * 3)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * 3)(*[b.size for b in bufs])
    yb, cb = g.y_blocks * 64, g.c_blocks * 64
    cy = np.full(3 * yb, 77, np.int16); cu = np.full(3 * cb, 77, np.int16); cv = cu.copy()
    q = np.full((3, 4, 64), 9, np.uint16)
    st = (C.c_int * 3)()
    for th in (1, 8):
        rc = capi.lib().ffhip_jpeg_entropy_batch(arr, lens, 3, th, C.byref(g), cy.ctypes.data, cu.ctypes.data, cv.ctypes.data, q.ctypes.data, st)
        assert rc == -22 and list(st) == [0, -22, 0]
        assert not cy[yb:2 * yb].any() and not cu[cb:2 * cb].any() and not cv[cb:2 * cb].any() and (q[1] == 1).all()
        assert np.array_equal(cy[:yb], cy[2 * yb:]) and cy[:yb].any()


@pytest.mark.skipif(not O.have_ref(), reason="needs oracle/_ref for the reference's bmpwriter")
def test_bmp_bytes_equal_reference_writer(tmp_path):
    R = O.ref()

    class Display(C.Structure):   # display/display.h:10-19
        _fields_ = [("name", C.c_char_p), ("width", C.c_int), ("height", C.c_int), ("private", C.c_void_p),
                    ("init", C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_int, C.c_int)), ("uninit", C.CFUNCTYPE(C.c_int)),
                    ("draw_pixels", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int))]
    bw = Display.in_dll(R, "bmp_writer")
    rng = np.random.default_rng(1)
    w, h = 40, 24
    px = rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
    title = str(tmp_path / "ref")
    bw.init(title.encode(), w, h)
    bw.draw_pixels(px.ctypes.data, 0, 0, w, h, 32, w * 4, 0)
    bw.uninit()
    mine = str(tmp_path / "mine.bmp")
    capi.check(capi.lib().ffhip_bmp_write(mine.encode(), px.ctypes.data, w, h, w * 4))
    assert open(mine, "rb").read() == open(title + ".bmp", "rb").read()
    # a padded pitch writes the same file
    padded = np.zeros((h, w + 8, 4), np.uint8)
    padded[:, :w] = px
    capi.check(capi.lib().ffhip_bmp_write(mine.encode(), padded.ctypes.data, w, h, (w + 8) * 4))
    assert open(mine, "rb").read() == open(title + ".bmp", "rb").read()


def test_front_end_under_sanitizers(tmp_path):
    """ASan + UBSan build of the C front end against seeded corruptions of the fixture files:
    bit flips, truncations, marker floods, header mutations (sanitizers run on the CPU build only)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_entropy")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "tools", "fuzz_entropy.c"),
                           os.path.join(root, "ffpic_amd", "csrc", "ffhip_entropy.c"), "-lpthread", "-o", exe])
    files = [os.path.join(GOLDEN, f) for f in FILES.values()]
    out = subprocess.run([exe, "400"] + files, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "decoded" in out.stdout


@pytest.mark.parametrize("blocks", [1, 7, 40])
def test_one_picture_over_threads_by_restart_interval(blocks):
    """ffhip_jpeg_entropy_decode_mt: the restart intervals of ONE picture shared out over host threads give the
    planes of the single-thread decode and of the fixture decoder, for any thread count"""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(blocks)
    yy, xx = np.mgrid[0:300, 0:420]
    img = np.stack([128 + 100 * np.sin(xx / 17.0), 128 + 90 * np.cos(yy / 13.0), (xx + yy) % 256], axis=2)
    img = np.clip(img + rng.normal(0, 8, img.shape), 0, 255).astype(np.uint8)
    bio = io.BytesIO()
    PIL.fromarray(img).save(bio, "JPEG", quality=88, subsampling=2, restart_marker_blocks=blocks)
    data = bio.getvalue()
    dec = jpeg_entropy.decode(data)
    L = capi.lib()
    g, _, _ = ops.jpeg_probe(data)
    buf = np.frombuffer(data, np.uint8)
    for th in (1, 2, 5, 8, 64):
        cy = np.full(g.y_blocks * 64, 77, np.int16)
        cu = np.full(g.c_blocks * 64, 77, np.int16)
        cv = np.full(g.c_blocks * 64, 77, np.int16)
        q = np.zeros((4, 64), np.uint16)
        rc = L.ffhip_jpeg_entropy_decode_mt(buf.ctypes.data, buf.size, C.byref(g), cy.ctypes.data, cu.ctypes.data, cv.ctypes.data, q.ctypes.data, th)
        assert rc == 0
        assert np.array_equal(cy, dec["coef"][0]) and np.array_equal(cu, dec["coef"][1]) and np.array_equal(cv, dec["coef"][2]), th
    # a missing marker is refused, not guessed around
    broken = bytearray(data)
    k = data.find(b"\xff\xd1")
    assert k > 0
    broken[k + 1] = 0x00
    bb = np.frombuffer(bytes(broken), np.uint8)
    assert L.ffhip_jpeg_entropy_decode_mt(bb.ctypes.data, bb.size, C.byref(g), cy.ctypes.data, cu.ctypes.data, cv.ctypes.data, q.ctypes.data, 4) != 0
