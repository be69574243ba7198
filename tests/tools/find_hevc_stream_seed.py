#!/usr/bin/env python3
"""Search for seeds of tests/hevc_bitstream.py streams that the REFERENCE's own parser decodes to a clean end: the slice data is seeded
random bytes (any byte string is a CABAC stream), and what makes a stream usable is that end_of_slice_segment_flag comes out 1 right
after the last coding tree unit and 0 before (coding/hevc.c:7007-7019) -- about one seed in 190 / survival, i.e. one in a few thousand for
a 1080p picture.  Build container only (needs oracle/_ref).
  find_hevc_stream_seed.py W H first_seed count [bytes_per_ctb] [jobs] [constrained_intra]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)


def attempt(w, h, seed, n_bytes, ci):
    import numpy as np
    import oracle_lib as O
    import hevc_bitstream as HB
    R = O.ref()
    R.ref_hevc_param_set_new.restype = C.c_void_p
    R.parse_nalu.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
    hps = R.ref_hevc_param_set_new()
    nals = HB.stream(w, h, seed, n_bytes, constrained_intra=ci)
    pix = np.zeros(w * (h + 64) * 4 + 4096, np.uint8)
    for n in nals[:3]:
        buf = np.frombuffer(n, np.uint8).copy()
        dummy = C.c_void_p(0)
        R.parse_nalu(buf.ctypes.data, buf.size, C.byref(dummy), hps)
    buf = np.frombuffer(nals[3], np.uint8).copy()
    pp = C.c_void_p(pix.ctypes.data)
    R.ref_hevc_record_begin()
    R.parse_nalu(buf.ctypes.data, buf.size, C.byref(pp), hps)       # exit(-1) inside when the stream does not end where the picture does
    info = (C.c_long * 8)()
    R.ref_hevc_record_end(info)
    rec = np.zeros(info[0], np.dtype([("x", "<i4"), ("y", "<i4"), ("log2", "<i4"), ("cidx", "<i4"), ("mode", "<i4"), ("flags", "<i4"), ("qp", "<i4"), ("rflags", "<i4"),
                                      ("level_off", "<i4"), ("pad", "<i4"), ("avail_top", "<u8"), ("avail_left", "<u8")]))
    lv = np.zeros(max(info[1], 1), np.int16); rs = np.zeros(max(info[1], 1), np.int16); pl = np.zeros(max(info[2], 1), np.int16)
    R.ref_hevc_record_fetch.argtypes = [C.c_void_p] * 4
    R.ref_hevc_record_fetch(rec.ctypes.data, lv.ctypes.data, rs.ctypes.data, pl.ctypes.data)
    luma = rec[rec["cidx"] == 0]
    ok = int((1 << (2 * luma["log2"].astype(np.int64))).sum()) == w * h and int(rec["pad"].sum()) == 0
    sys.stdout.flush()
    os._exit(0 if ok else 3)


if __name__ == "__main__":
    if sys.argv[1] == "--try":
        attempt(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
    w, h, first, count = (int(x) for x in sys.argv[1:5])
    per_ctb = int(sys.argv[5]) if len(sys.argv) > 5 else 2600
    jobs = int(sys.argv[6]) if len(sys.argv) > 6 else 6
    ci = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    n_bytes = per_ctb * ((w + 63) // 64) * ((h + 63) // 64)
    from concurrent.futures import ThreadPoolExecutor

    def run(seed):
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--try", str(w), str(h), str(seed), str(n_bytes), str(ci)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return seed, rc
    found = []
    with ThreadPoolExecutor(jobs) as ex:
        for seed, rc in ex.map(run, range(first, first + count)):
            if rc == 0:
                found.append(seed)
                print("seed", seed, "n_bytes", n_bytes, flush=True)
    print("found", found, "of", count)
