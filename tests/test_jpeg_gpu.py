"""GPU parity tests for the JPEG hot path, through the C ABI of libffpic_hip.so.
Bit-exact (integer/byte work) against the golden vectors and the CPU oracle."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O
from ffpic_amd import capi, ops, synth
from test_oracle_golden import FILES, GRID_TAGS, decode_fixture

pytestmark = pytest.mark.gpu


def to_capi(g):
    return capi.jpeg_geom(g.mcu_cols, g.mcu_rows, g.ncomp, g.h, g.v, tuple(g.qt_id))


def gpu_recon(geom, n, cy, cu, cv, quant):
    return ops.jpeg_recon_batch_host(to_capi(geom), n, cy, cu, cv, quant)


def test_device_is_gfx950():
    L = capi.require_device()
    assert L.ffhip_arch_name() == b"gfx950"


@pytest.mark.parametrize("tag", list(GRID_TAGS))
def test_golden_grids(golden, tag):
    g = golden("jpeg_grids.npz")
    cols, rows, nc, h, v = GRID_TAGS[tag]
    geom = O.make_geom(cols, rows, nc, h, v)
    cy, cu, cv = synth.coef_batch(1, cols, rows, nc, h, v)
    out = gpu_recon(geom, 1, cy, cu, cv, g["quant"])[0]
    assert np.array_equal(out, g[f"{tag}_bgra"])


def test_golden_adversarial(golden):
    """full-range int16 levels x full-range uint16 quant: every int16 truncation point
    and the mod-2^32 accumulation of the dot products"""
    g = golden("jpeg_grids.npz")
    geom = O.make_geom(*[int(x) for x in g["adv_geom"][:5]])
    out = gpu_recon(geom, 1, g["adv_cy"], g["adv_cu"], g["adv_cv"], g["adv_quant"])[0]
    assert np.array_equal(out, g["adv_bgra"])
    for tag in ("adv411", "adv114"):          # the same blocks as 4:1:1 (h = 4) and as its transpose (v = 4)
        geom = O.make_geom(*[int(x) for x in g[f"{tag}_geom"][:5]])
        out = gpu_recon(geom, 1, g["adv_cy"], g["adv_cu"], g["adv_cv"], g["adv_quant"])[0]
        assert np.array_equal(out, g[f"{tag}_bgra"]), tag


@pytest.mark.parametrize("tag", list(FILES))
def test_golden_files(golden, tag):
    g = golden("jpeg_files.npz")
    dec, geom = decode_fixture(tag)
    out = gpu_recon(geom, 1, dec["coef"][0], dec["coef"][1], dec["coef"][2], dec["quant"])[0]
    H, W = [int(x) for x in g[f"{tag}_shape"][:2]]
    out = out[:H, :W]
    if int(g[f"{tag}_last_mcu_exact"]):
        assert hashlib.sha256(out.tobytes()).digest() == g[f"{tag}_sha256"].tobytes()
    else:
        keep = np.ones((H, W), bool)
        keep[(geom.mcu_rows - 1) * 8 * geom.v:, (geom.mcu_cols - 1) * 8 * geom.h:] = False
        assert np.array_equal(out[keep], g[f"{tag}_bgra"][keep])


@pytest.mark.parametrize("cols,rows,n", [(1, 1, 1), (2, 1, 3), (3, 2, 2), (4, 4, 1), (5, 1, 4), (13, 7, 3),
                                         (40, 30, 2), (120, 68, 1)])
def test_420_vs_oracle(cols, rows, n):
    """ragged MCU counts (tail quads), several images per batch, per-image quant tables"""
    geom = O.make_geom(cols, rows)
    rng = np.random.default_rng(cols * 100 + rows)
    q = np.stack([synth.quant_tables(int(rng.integers(30, 96))) for _ in range(n)])
    ys, us, vs = [], [], []
    for i in range(n):
        y, u, v = synth.coef_image(i, cols, rows, quant=q[i])
        ys.append(y); us.append(u); vs.append(v)
    cy, cu, cv = (np.ascontiguousarray(np.concatenate(a).reshape(-1)) for a in (ys, us, vs))
    exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n, n_threads=4)
    out = gpu_recon(geom, n, cy, cu, cv, q)
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("nc,h,v", [(3, 1, 1), (3, 2, 1), (3, 1, 2), (1, 1, 1), (3, 4, 1), (3, 1, 4), (3, 3, 1), (3, 1, 3),
                                    (1, 4, 1), (1, 1, 4)])
def test_other_geometries_vs_oracle(nc, h, v):
    geom = O.make_geom(11, 6, nc, h, v)
    q = synth.quant_tables(70)
    cy, cu, cv = synth.coef_batch(3, 11, 6, nc, h, v, quant=q)
    exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=3)
    assert np.array_equal(gpu_recon(geom, 3, cy, cu, cv, q), exp)


def test_adversarial_random_vs_oracle():
    rng = np.random.default_rng(2024)
    geom = O.make_geom(16, 8)
    blocks = synth.adversarial_blocks(rng, 16 * 8 * 6)
    rng.shuffle(blocks)
    cy = np.ascontiguousarray(blocks[:512].reshape(-1))
    cu = np.ascontiguousarray(blocks[512:640].reshape(-1))
    cv = np.ascontiguousarray(blocks[640:].reshape(-1))
    q = rng.integers(1, 65536, size=(4, 64)).astype(np.uint16)
    exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q)
    assert np.array_equal(gpu_recon(geom, 1, cy, cu, cv, q), exp)


def test_exact_integer_green_branch(golden):
    """Force the rare fp64 branch of the fused kernel: flat (DC-only) blocks reproduce
    any sample value s = 128 + dc/8, so every MCU gets a chroma pair with
    215*uu + 381*vv == 0 (mod 1000) and luma values around the clamp range."""
    tri = golden("color_triples.npz")["yuv"]
    uu, vv = tri[:, 1].astype(np.int64) - 128, tri[:, 2].astype(np.int64) - 128
    sel = ((215 * uu + 381 * vv) % 1000 == 0) & (tri[:, 1] >= 0) & (tri[:, 1] < 4000) & (tri[:, 2] >= 0) & \
          (tri[:, 2] < 4000) & ((uu != 0) | (vv != 0))
    pairs = np.unique(tri[sel][:, 1:3], axis=0)
    assert len(pairs) > 1000
    cols, rows = 48, 40
    pairs = pairs[: cols * rows]
    n = len(pairs)
    rng = np.random.default_rng(3)
    geom = O.make_geom(cols, rows)
    cy = np.zeros((cols * rows * 4, 64), np.int16)
    cu = np.zeros((cols * rows, 64), np.int16)
    cv = np.zeros((cols * rows, 64), np.int16)
    cu[:n, 0] = (pairs[:, 0].astype(np.int64) - 128) * 8
    cv[:n, 0] = (pairs[:, 1].astype(np.int64) - 128) * 8
    # luma such that G lands in and around [0, 255]: yy ~ (215 uu + 381 vv)/1000 + U(-3, 258)
    base = (215 * (pairs[:, 0].astype(np.int64) - 128) + 381 * (pairs[:, 1].astype(np.int64) - 128)) // 1000
    yy = np.clip(base[:, None] + rng.integers(-3, 259, size=(n, 4)), 0, 4000)
    cy[: n * 4, 0] = ((yy - 128) * 8).reshape(-1)
    q = np.ones((4, 64), np.uint16)
    args = (np.ascontiguousarray(cy.reshape(-1)), np.ascontiguousarray(cu.reshape(-1)),
            np.ascontiguousarray(cv.reshape(-1)), q)
    exp = O.oracle_jpeg_recon(geom, *args)
    out = gpu_recon(geom, 1, *args)
    assert np.array_equal(out, exp)
    # the branch really was exercised: the integer form alone would be wrong somewhere
    s = 215 * (pairs[:, 0].astype(np.int64) - 128) + 381 * (pairs[:, 1].astype(np.int64) - 128)
    g_int = np.clip(yy - s[:, None] // 1000, 0, 255)
    g_ref = exp[0].reshape(rows, 16, cols, 16, 4)[:, ::8, :, ::8, 1].transpose(0, 2, 1, 3).reshape(-1, 4)[:n]
    assert (g_int != g_ref).any()


def test_device_pointer_api_and_full_size_properties():
    """BASELINE config 2 geometry (1920x1088 coded) on device memory: a few images are
    checked bit-for-bit against the oracle, and the whole batch through size-independent
    properties: every image's output equals its output when decoded alone (checksum of
    checksums), the output does not depend on the batch position, and bytes beyond the
    pitch/stride windows are untouched."""
    torch = pytest.importorskip("torch")
    L = capi.require_device()
    dev = torch.device("cuda:0")
    cols, rows, n_unique, n = 120, 68, 4, 24
    geom = O.make_geom(cols, rows)
    cg = to_capi(geom)
    q = synth.quant_tables()
    cy, cu, cv = synth.coef_batch(n_unique, cols, rows)
    reps = n // n_unique
    t_y = torch.from_numpy(cy).to(dev).repeat(reps)
    t_u = torch.from_numpy(cu).to(dev).repeat(reps)
    t_v = torch.from_numpy(cv).to(dev).repeat(reps)
    t_q = torch.from_numpy(q.astype(np.int16)).to(dev)
    H, W = geom.height, geom.width
    pitch = W * 4 + 256                 # padded pitch
    stride = pitch * H + 4096           # padded image stride
    out = torch.full((n * stride,), 0xA5, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ops.jpeg_recon_batch(cg, n, t_y.data_ptr(), t_u.data_ptr(), t_v.data_ptr(), t_q.data_ptr(), 0, out.data_ptr(),
                         pitch, stride, None, 0, st)
    torch.cuda.synchronize()
    o = out.view(n, stride)
    img = o[:, : pitch * H].view(n, H, pitch)
    pix = img[:, :, : W * 4]
    assert bool((img[:, :, W * 4:] == 0xA5).all()) and bool((o[:, pitch * H:] == 0xA5).all())
    exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n_unique, n_threads=4)
    got = pix[:n_unique].cpu().numpy().reshape(n_unique, H, W, 4)
    assert np.array_equal(got, exp)
    sums = pix.reshape(n, -1).to(torch.int64).mul(torch.arange(1, 1 + H * W * 4, device=dev) % 251).sum(dim=1)
    assert bool((sums.view(reps, n_unique) == sums[:n_unique]).all())


def test_host_and_device_paths_agree_and_empty_batch():
    L = capi.require_device()
    geom = O.make_geom(8, 3)
    cg = to_capi(geom)
    assert L.ffhip_jpeg_recon_batch_host(C.byref(cg), 0, None, None, None, None, 0, None, 0, 0) == 0
    q = synth.quant_tables()
    cy, cu, cv = synth.coef_batch(2, 8, 3)
    a = gpu_recon(geom, 2, cy, cu, cv, q)
    # same data through malloc/memcpy helpers of the C ABI (no torch)
    H, W = geom.height, geom.width
    bufs = []
    for arr in (cy, cu, cv, q):
        d = L.ffhip_malloc(arr.nbytes)
        assert d
        capi.check(L.ffhip_memcpy_h2d(d, arr.ctypes.data, arr.nbytes, None))
        bufs.append(d)
    dout = L.ffhip_malloc(2 * H * W * 4)
    st = L.ffhip_stream_create()
    e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
    capi.check(L.ffhip_stream_sync(None))
    capi.check(L.ffhip_event_record(e0, st))
    capi.check(L.ffhip_jpeg_recon_batch(C.byref(cg), 2, bufs[0], bufs[1], bufs[2], bufs[3], 0, dout, W * 4,
                                        H * W * 4, None, 0, st))
    capi.check(L.ffhip_event_record(e1, st))
    assert L.ffhip_event_elapsed_ms(e0, e1) >= 0.0
    b = np.empty_like(a)
    capi.check(L.ffhip_memcpy_d2h(b.ctypes.data, dout, b.nbytes, st))
    capi.check(L.ffhip_stream_sync(st))
    assert np.array_equal(a, b)
    for d in bufs + [dout]:
        L.ffhip_free(d)
    L.ffhip_event_destroy(e0); L.ffhip_event_destroy(e1); L.ffhip_stream_destroy(st)
    # misaligned / undersized arguments are refused, not "fixed up"
    assert L.ffhip_jpeg_recon_batch(C.byref(cg), 1, 16, 16, 16, 16, 0, 16, W * 4 - 16, 0, None, 0, None) == -22
    assert L.ffhip_jpeg_recon_batch(C.byref(cg), 1, 8, 16, 16, 16, 0, 16, W * 4, 0, None, 0, None) == -22


@pytest.mark.parametrize("tag", list(FILES))
def test_end_to_end_file_to_bgra(golden, tag, tmp_path):
    """transbmp-equivalent: .jpg bytes -> C entropy front end (host) -> fused reconstruction (GPU)
    -> BMP, against what the reference's own loader decoded from the same file"""
    g = golden("jpeg_files.npz")
    data = open(os.path.join(os.path.dirname(__file__), "golden", FILES[tag]), "rb").read()
    geom, out = ops.decode_jpeg_files([data, data], n_threads=2)
    H, W = [int(x) for x in g[f"{tag}_shape"][:2]]
    assert np.array_equal(out[0], out[1])
    img = out[0][:H, :W]
    if int(g[f"{tag}_last_mcu_exact"]):
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest() == g[f"{tag}_sha256"].tobytes()
    else:
        keep = np.ones((H, W), bool)
        keep[(geom.mcu_rows - 1) * 8 * geom.v:, (geom.mcu_cols - 1) * 8 * geom.h:] = False
        assert np.array_equal(img[keep], g[f"{tag}_bgra"][keep])
    path = str(tmp_path / "out.bmp")
    full = np.ascontiguousarray(out[0])
    capi.check(capi.lib().ffhip_bmp_write(path.encode(), full.ctypes.data, W, H, full.shape[1] * 4))
    raw = open(path, "rb").read()
    assert len(raw) == 54 + W * H * 4 and raw[:2] == b"BM" and raw[54:54 + 16] == img[0, :4].tobytes()


def test_c_host_program_transbmp(golden, tmp_path):
    """examples/transbmp_hip.c: the whole file -> BMP flow from plain C through the C ABI."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "transbmp_hip")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "transbmp_hip.c"), "-L" + os.path.join(root, "ffpic_amd"),
                           "-lffpic_hip", "-Wl,-rpath," + os.path.join(root, "ffpic_amd"), "-o", exe])
    src = str(tmp_path / "pic.jpg")
    shutil.copy(os.path.join(root, "tests", "golden", FILES["q85_420"]), src)
    out = subprocess.run([exe, src], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    bmp = open(src + " (640 * 480).bmp", "rb").read()
    assert len(bmp) == 54 + 640 * 480 * 4
    g = golden("jpeg_files.npz")
    assert hashlib.sha256(bmp[54:]).digest() == g["q85_420_sha256"].tobytes()   # == the reference's decode of the file


STRIP_LAYOUTS = [(3, 1, 1), (3, 2, 1), (3, 1, 2), (1, 1, 1), (3, 4, 1), (3, 1, 4)]


@pytest.mark.parametrize("nc,h,v", STRIP_LAYOUTS + [(1, 2, 2), (3, 3, 1), (3, 1, 3)])
@pytest.mark.parametrize("cols,rows,n", [(1, 1, 1), (7, 1, 2), (8, 2, 1), (9, 3, 2), (17, 2, 1), (33, 1, 3)])
def test_strip_kernel_sizes(nc, h, v, cols, rows, n):
    """k_jpeg_fused_strip (and, for grey with 2x2 blocks per MCU, the two-pass path): full and ragged strips"""
    geom = O.make_geom(cols, rows, nc, h, v)
    q = synth.quant_tables(60 + cols)
    cy, cu, cv = synth.coef_batch(n, cols, rows, nc, h, v, quant=q, first=cols * 10 + rows)
    exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n)
    assert np.array_equal(gpu_recon(geom, n, cy, cu, cv, q), exp)


@pytest.mark.parametrize("nc,h,v", STRIP_LAYOUTS)
def test_strip_kernel_adversarial(nc, h, v):
    """int16-wrapping coefficients and arbitrary quant factors: samples anywhere in the IDCT's range"""
    rng = np.random.default_rng(77 + h + 2 * v + nc)
    cols, rows = 12, 5
    geom = O.make_geom(cols, rows, nc, h, v)
    nby = cols * rows * (h * v if nc == 3 else 1)
    blocks = synth.adversarial_blocks(rng, nby + 2 * cols * rows)
    rng.shuffle(blocks)
    cy = np.ascontiguousarray(blocks[:nby].reshape(-1))
    cu = np.ascontiguousarray(blocks[nby:nby + cols * rows].reshape(-1)) if nc == 3 else None
    cv = np.ascontiguousarray(blocks[nby + cols * rows:].reshape(-1)) if nc == 3 else None
    q = rng.integers(1, 65536, size=(4, 64)).astype(np.uint16)
    exp = O.oracle_jpeg_recon(geom, cy, cu, cv, q)
    assert np.array_equal(gpu_recon(geom, 1, cy, cu, cv, q), exp)


@pytest.mark.parametrize("h,v", [(1, 1), (2, 1), (1, 2), (4, 1), (1, 4)])
def test_strip_kernel_exact_integer_green(golden, h, v):
    """flat blocks carrying chroma pairs with 215 uu + 381 vv == 0 (mod 1000): the fp64 branch of the strip kernel"""
    tri = golden("color_triples.npz")["yuv"]
    uu, vv = tri[:, 1].astype(np.int64) - 128, tri[:, 2].astype(np.int64) - 128
    sel = ((215 * uu + 381 * vv) % 1000 == 0) & (tri[:, 1] >= 0) & (tri[:, 1] < 4000) & (tri[:, 2] >= 0) & \
          (tri[:, 2] < 4000) & ((uu != 0) | (vv != 0))
    pairs = np.unique(tri[sel][:, 1:3], axis=0)
    cols, rows = 40, 30
    pairs = pairs[: cols * rows]
    n = len(pairs)
    assert n == cols * rows
    rng = np.random.default_rng(5)
    geom = O.make_geom(cols, rows, 3, h, v)
    bpm = h * v
    cy = np.zeros((n * bpm, 64), np.int16)
    cu = np.zeros((n, 64), np.int16)
    cv = np.zeros((n, 64), np.int16)
    cu[:, 0] = (pairs[:, 0].astype(np.int64) - 128) * 8
    cv[:, 0] = (pairs[:, 1].astype(np.int64) - 128) * 8
    base = (215 * (pairs[:, 0].astype(np.int64) - 128) + 381 * (pairs[:, 1].astype(np.int64) - 128)) // 1000
    yy = np.clip(base[:, None] + rng.integers(-3, 259, size=(n, bpm)), 0, 4000)
    cy[:, 0] = ((yy - 128) * 8).reshape(-1)
    q = np.ones((4, 64), np.uint16)
    args = (np.ascontiguousarray(cy.reshape(-1)), np.ascontiguousarray(cu.reshape(-1)), np.ascontiguousarray(cv.reshape(-1)), q)
    exp = O.oracle_jpeg_recon(geom, *args)
    got = gpu_recon(geom, 1, *args)
    assert np.array_equal(got, exp)
    # the branch really is exercised: the integer form alone would differ somewhere
    g_int = np.clip(yy[:, 0] + (-(215 * (pairs[:, 0].astype(np.int64) - 128) + 381 * (pairs[:, 1].astype(np.int64) - 128))) // 1000, 0, 255)
    assert (g_int != exp.reshape(rows, 8 * v, cols, 8 * h, 4)[:, 0, :, 0, 1].reshape(-1)).any()


@pytest.mark.parametrize("tag", list(FILES))
@pytest.mark.parametrize("n,chunk,threads", [(1, 0, 1), (5, 2, 3), (9, 4, 16)])
def test_files_to_pixels_pipeline(tag, n, chunk, threads):
    """ffhip_jpeg_decode_files: the double-buffered host-entropy / GPU-reconstruction pipeline gives the bytes of the
    plain two-step path (entropy batch, then the host-buffer reconstruction) for every fixture file"""
    data = open(os.path.join(os.path.dirname(__file__), "golden", FILES[tag]), "rb").read()
    files = [data] * n
    g, ref = ops.decode_jpeg_files(files[:1], n_threads=1)
    g2, out = ops.jpeg_decode_files(files, n_threads=threads, chunk=chunk)
    assert (g2.mcu_cols, g2.mcu_rows, g2.ncomp, g2.h, g2.v) == (g.mcu_cols, g.mcu_rows, g.ncomp, g.h, g.v)
    ref = np.asarray(ref[0]).reshape(g.height, g.width, 4)
    for i in range(n):
        assert np.array_equal(out[i], ref), (tag, i)
    # a pinned destination takes the device copy directly: same bytes
    pin = ops.PinnedArray((n, g.height, g.width, 4))
    pin.array[:] = 0
    ops.jpeg_decode_files(files, n_threads=threads, chunk=chunk, out=pin.array)
    assert np.array_equal(pin.array, out)


@pytest.mark.parametrize("tag", list(FILES))
def test_files_to_device_pixels(tag):
    """ffhip_jpeg_decode_files_device (entropy on the device for the DRI fixture, on host threads for the others)
    leaves in device memory what the host-destination pipeline returns"""
    data = open(os.path.join(os.path.dirname(__file__), "golden", FILES[tag]), "rb").read()
    files = [data] * 3
    g, ref = ops.jpeg_decode_files(files, n_threads=2)
    g2, out, _ = ops.jpeg_decode_files_device(files, n_threads=2)
    assert np.array_equal(out, ref)


def test_pipeline_with_device_entropy_forced_on_plain_files(monkeypatch):
    """FFHIP_JPEG_GPU_ENTROPY=1: files without restart markers through the device decoder (one lane per file), same bytes"""
    data = open(os.path.join(os.path.dirname(__file__), "golden", FILES["q85_420"]), "rb").read()
    files = [data] * 6
    monkeypatch.setenv("FFHIP_JPEG_GPU_ENTROPY", "0"); capi.reload_env()
    _, ref = ops.jpeg_decode_files(files, n_threads=2, chunk=4)
    monkeypatch.setenv("FFHIP_JPEG_GPU_ENTROPY", "1"); capi.reload_env()
    _, out = ops.jpeg_decode_files(files, n_threads=2, chunk=4)
    _, dev, _ = ops.jpeg_decode_files_device(files, n_threads=2)
    assert np.array_equal(out, ref) and np.array_equal(dev, ref)


def test_shutdown_releases_and_rebinds():
    """ffhip_shutdown frees what the library keeps between calls; the next call starts from scratch and is still exact"""
    L = capi.require_device()
    data = open(os.path.join(os.path.dirname(__file__), "golden", FILES["q85_420_dri"]), "rb").read()
    _, a = ops.jpeg_decode_files([data] * 3, n_threads=2)
    tus, res = synth.hevc_intra_tus(128, 128, 5)
    ya = ops.hevc_intra_recon(tus, res, 128, 128)[0]
    L.ffhip_shutdown()
    L.ffhip_shutdown()                      # idempotent
    capi.require_device()
    _, b = ops.jpeg_decode_files([data] * 3, n_threads=2)
    yb = ops.hevc_intra_recon(tus, res, 128, 128)[0]
    assert np.array_equal(a, b) and np.array_equal(ya, yb)


@pytest.mark.parametrize("cols,rows,pad", [(13, 7, 0), (40, 30, 1024), (5, 1, 0)])
def test_pattern_calibration_touches_what_the_kernel_touches(cols, rows, pad):
    """ffhip_jpeg_pattern_calibrate (bench.py's roofline.pattern_GBps): the fused 4:2:0 kernel's loads and stores without the arithmetic.  It must write
    exactly the bytes the real kernel writes -- every pixel of the coded picture, nothing in the pitch padding, nothing of a ragged quad's missing MCUs --
    and leave the next real launch bit-exact; other layouts are refused."""
    import ctypes as C
    L = capi.require_device()
    geom = O.make_geom(cols, rows)
    cg = to_capi(geom)
    n = 2
    q = synth.quant_tables(80)
    cy, cu, cv = synth.coef_batch(n, cols, rows, quant=q)
    W, H = cols * 16, rows * 16
    pitch = W * 4 + pad
    stride = pitch * H
    dy, du, dv = ops.DeviceBuffer(cy), ops.DeviceBuffer(cu), ops.DeviceBuffer(cv)
    dq = ops.DeviceBuffer(np.ascontiguousarray(q.astype(np.uint16)))
    out = ops.DeviceBuffer(nbytes=n * stride + 4096)
    capi.check(L.ffhip_memset(out.ptr, 0xA5, n * stride + 4096, None))
    capi.setenv("FFHIP_JPEG_PATTERN_SLEEP", "3" if pad else None)      # (diagnostics: the wave idles between its loads and its stores)
    try:
        capi.check(L.ffhip_jpeg_pattern_calibrate(C.byref(cg), n, dy.ptr, du.ptr, dv.ptr, dq.ptr, 0, out.ptr, pitch, stride, None))
        capi.check(L.ffhip_stream_sync(None))
    finally:
        capi.setenv("FFHIP_JPEG_PATTERN_SLEEP", None)
    raw = out.to_host((n * stride + 4096,), np.uint8)
    assert (raw[n * stride:] == 0xA5).all()                               # nothing behind the last picture
    img = raw[:n * stride].reshape(n, H, pitch)
    if pad:
        assert (img[:, :, W * 4:] == 0xA5).all()                           # nothing in the padding of a row
    # the words it stores are XORs of coefficient words + a small counter: with random coefficients practically none equals the fill pattern
    touched = (np.ascontiguousarray(img[:, :, :W * 4]).reshape(n, H, W, 4).view(np.uint32)[..., 0] != 0xA5A5A5A5)
    assert touched.mean() > 0.99
    capi.check(L.ffhip_jpeg_recon_batch(C.byref(cg), n, dy.ptr, du.ptr, dv.ptr, dq.ptr, 0, out.ptr, pitch, stride, None, 0, None))
    capi.check(L.ffhip_stream_sync(None))
    got = out.to_host((n * stride + 4096,), np.uint8)[:n * stride].reshape(n, H, pitch)[:, :, :W * 4].reshape(n, H, W, 4)
    assert np.array_equal(got, O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n))
    g31 = to_capi(O.make_geom(cols, rows, 3, 3, 1))     # a two-pass geometry has no fused kernel, hence no twin
    assert L.ffhip_jpeg_pattern_calibrate(C.byref(g31), 1, dy.ptr, du.ptr, dv.ptr, dq.ptr, 0, out.ptr, cols * 24 * 4, cols * 24 * 4 * rows * 8, None) == capi.FFHIP_EINVAL


@pytest.mark.parametrize("nc,h,v", [(3, 1, 1), (3, 2, 1), (3, 1, 2), (1, 1, 1), (3, 4, 1), (3, 1, 4)])
def test_pattern_calibration_of_the_strip_layouts(nc, h, v):
    """the arithmetic-free twins of k_jpeg_fused_strip: every pixel of the coded picture written (ragged strip ends included), nothing beyond it, and the
    next real launch bit-exact"""
    import ctypes as C
    L = capi.require_device()
    cols, rows, n = 11, 6, 2
    geom = O.make_geom(cols, rows, nc, h, v)
    cg = to_capi(geom)
    q = synth.quant_tables(70)
    cy, cu, cv = synth.coef_batch(n, cols, rows, nc, h, v, quant=q)
    W, H = cols * 8 * h, rows * 8 * v
    pitch = (W * 4 + 15) // 16 * 16 + 64
    stride = pitch * H
    dy = ops.DeviceBuffer(cy)
    du, dv = (ops.DeviceBuffer(cu), ops.DeviceBuffer(cv)) if nc == 3 else (None, None)
    dq = ops.DeviceBuffer(np.ascontiguousarray(q.astype(np.uint16)))
    out = ops.DeviceBuffer(nbytes=n * stride + 4096)
    capi.check(L.ffhip_memset(out.ptr, 0xA5, n * stride + 4096, None))
    up, vp = (du.ptr, dv.ptr) if nc == 3 else (None, None)
    capi.check(L.ffhip_jpeg_pattern_calibrate(C.byref(cg), n, dy.ptr, up, vp, dq.ptr, 0, out.ptr, pitch, stride, None))
    capi.check(L.ffhip_stream_sync(None))
    raw = out.to_host((n * stride + 4096,), np.uint8)
    assert (raw[n * stride:] == 0xA5).all()
    img = raw[:n * stride].reshape(n, H, pitch)
    assert (img[:, :, W * 4:] == 0xA5).all()
    touched = np.ascontiguousarray(img[:, :, :W * 4]).reshape(n, H, W, 4).view(np.uint32)[..., 0] != 0xA5A5A5A5
    assert touched.mean() > 0.99
    capi.check(L.ffhip_jpeg_recon_batch(C.byref(cg), n, dy.ptr, up, vp, dq.ptr, 0, out.ptr, pitch, stride, None, 0, None))
    capi.check(L.ffhip_stream_sync(None))
    got = np.ascontiguousarray(out.to_host((n * stride + 4096,), np.uint8)[:n * stride].reshape(n, H, pitch)[:, :, :W * 4]).reshape(n, H, W, 4)
    assert np.array_equal(got, O.oracle_jpeg_recon(geom, cy, cu, cv, q, n_images=n))
