"""Host-side scheduling of the HEVC intra stage (no GPU): the group plan ffhip_hevc_intra_recon builds
must hand out tickets so that nothing ever waits for a group with a larger ticket -- that order is what
makes the single-launch form deadlock-free."""
import numpy as np
import pytest

from ffpic_amd import capi, synth


def plan(tus, w, h, cw, ch, wl=0):
    L = capi.lib()
    tk = np.zeros(len(tus), np.uint32)
    wt = np.zeros(len(tus), np.uint32)
    st = np.zeros(4, np.int32)
    rc = L.ffhip_hevc_intra_plan(tus.ctypes.data, len(tus), w, h, cw, ch, wl, tk.ctypes.data, wt.ctypes.data, st.ctypes.data)
    return rc, tk, wt, st


def neighbour_owners(tus, dims):
    """for every TU the set of TUs whose samples its availability masks point at (4x4 granularity)"""
    owner = [np.full((h // 4, w // 4), -1, np.int64) for (w, h) in dims]
    out = []
    for i, t in enumerate(tus):
        c, n, x, y = int(t["cidx"]), 1 << int(t["log2_size"]), int(t["x"]), int(t["y"])
        o, deps = owner[c], set()
        if int(t["flags"]) & 1:
            deps.add(int(o[(y - 1) // 4, (x - 1) // 4]))
        at, al = int(t["avail_top"]), int(t["avail_left"])
        for k in range(0, 2 * n, 4):
            if (at >> k) & 15:
                deps.add(int(o[(y - 1) // 4, (x + k) // 4]))
            if (al >> k) & 15:
                deps.add(int(o[(y + k) // 4, (x - 1) // 4]))
        deps.discard(-1)
        out.append(deps)
        o[y // 4:(y + n) // 4, x // 4:(x + n) // 4] = i
    return out


@pytest.mark.parametrize("w,h,seed,ctb,c444,wl", [(256, 192, 1, 64, False, 0), (256, 128, 2, 64, False, 6), (256, 128, 3, 64, False, 3),
                                                 (192, 128, 4, 32, False, 6), (128, 128, 5, 64, True, 0), (128, 64, 6, 16, False, 6)])
def test_ticket_order_is_deadlock_free(w, h, seed, ctb, c444, wl):
    tus, _ = synth.hevc_intra_tus(w, h, seed, ctb=ctb, adversarial_masks=True, chroma_444=c444)
    cw, ch = (w, h) if c444 else (w // 2, h // 2)
    rc, tk, wt, st = plan(tus, w, h, cw, ch, wl)
    assert rc == 0
    groups, used_wl, n_wait, _ = [int(v) for v in st]
    assert groups == len(np.unique(tk)) and int(tk.max()) == groups - 1
    assert used_wl <= (wl or 5) and (1 << used_wl) <= max(ctb, 8)     # shrunk to the coding tree block where needed
    deps = neighbour_owners(tus, [(w, h), (cw, ch), (cw, ch)])
    waits = 0
    for i, d in enumerate(deps):
        other = {j for j in d if tk[j] != tk[i]}
        assert all(tk[j] < tk[i] for j in other), i               # never wait for a later ticket
        assert all(j < i for j in d)                              # and inside a group: decode order
        assert wt[i] == len(other), i
        waits += len(other)
    assert waits == n_wait or (waits == 0 and n_wait == 1)


def test_plan_rejects_a_list_no_window_fits():
    """two TUs that each sit in the other's past: whatever the window, some group would wait for a later one"""
    t = np.zeros(3, dtype=synth.HEVC_TU_DTYPE)
    # decode order A(0,0) B(64,0) C(0,0)-window again at (8,0) reading B's column: group(A,C) appears first but waits for B
    t["x"] = [0, 64, 60]
    t["y"] = [0, 0, 8]
    t["log2_size"] = 2
    t["avail_top"] = [0, 0, 0xF0]          # C reads (64..67, 7): B's bottom row
    rc, *_ = plan(t, 128, 64, 0, 0, 6)
    assert rc == 0 or rc == capi.FFHIP_EINVAL    # either a smaller window separates them, or the planner refuses
    rc3, tk, _, _ = plan(t, 128, 64, 0, 0, 3)
    if rc3 == 0:
        assert tk[1] < tk[2]


def check_order(tus, dims, tk, wt):
    deps = neighbour_owners(tus, dims)
    for i, d in enumerate(deps):
        other = {j for j in d if tk[j] != tk[i]}
        assert all(tk[j] < tk[i] for j in other), i
        assert wt[i] == len(other), i


@pytest.mark.parametrize("w,h,seed,kw", [(256, 192, 1, dict(tu_mix="c5")), (256, 128, 2, dict(adversarial_masks=True)),
                                         (128, 128, 5, dict(chroma_444=True)), (192, 128, 7, dict(min_tu=8))])
def test_reference_order_plans_like_plane_order(w, h, seed, kw):
    """The reference decodes per coding unit: luma tree, Cb, Cr (coding/hevc.c:5013-5180).  A run, a window visit and a cell visit are
    defined on each plane's OWN subsequence of the list, so the list in that order gets the same windows, groups and wait entries as the
    list with each coding tree block's planes one after the other -- group for group (the same TUs share a ticket group)."""
    a, _ = synth.hevc_intra_tus(w, h, seed, **kw)
    b, _ = synth.hevc_intra_tus(w, h, seed, order="reference", **kw)
    assert (b["cidx"][1:] != b["cidx"][:-1]).sum() > (a["cidx"][1:] != a["cidx"][:-1]).sum()
    cw, ch = (w, h) if kw.get("chroma_444") else (w // 2, h // 2)
    rca, tka, wta, sta = plan(a, w, h, cw, ch, 6)
    rcb, tkb, wtb, stb = plan(b, w, h, cw, ch, 6)
    assert rca == 0 and rcb == 0
    assert list(sta) == list(stb) and int(stb[1]) == 6
    check_order(b, [(w, h), (cw, ch), (cw, ch)], tkb, wtb)
    key = lambda t: (int(t["cidx"]), int(t["y"]), int(t["x"]))
    ga = {}
    for i, t in enumerate(a):
        ga.setdefault(int(tka[i]), set()).add(key(t))
    gb = {}
    for i, t in enumerate(b):
        gb.setdefault(int(tkb[i]), set()).add(key(t))
    assert sorted(map(sorted, ga.values())) == sorted(map(sorted, gb.values()))


def test_by_plane_switch_off_still_plans(monkeypatch):
    """FFHIP_HEVC_BY_PLANE=0 is the rule until round 5 (runs of the list as it is).  The HOST planner takes groups in order of first
    appearance whether or not they are contiguous runs, so it still plans the reference's order at 64x64 -- with decode-order tickets; it is
    the device planner that needs contiguous runs (tests/test_hevc_intra_gpu.py::test_reference_order_takes_the_wavefront_schedule)"""
    b, _ = synth.hevc_intra_tus(256, 192, 1, tu_mix="c5", order="reference")
    monkeypatch.setenv("FFHIP_HEVC_BY_PLANE", "0"); capi.reload_env()
    rc, tk, wt, st = plan(b, 256, 192, 128, 96, 6)
    monkeypatch.delenv("FFHIP_HEVC_BY_PLANE"); capi.reload_env()
    assert rc == 0
    check_order(b, [(256, 192), (128, 96), (128, 96)], tk, wt)


@pytest.mark.parametrize("name,tags", [("hevc_file.npz", "abcdef"), ("hevc_file_1080p.npz", "g")])
def test_lists_the_reference_recorded_take_the_64_window(golden, name, tags):
    """the seven TU lists the reference's own decoder recorded (plane-major per coding unit): 64x64 windows, one group per window and plane"""
    g = golden(name)
    for tag in tags:
        w, h = int(g[f"{tag}_dims"][0]), int(g[f"{tag}_dims"][1])
        tus = np.ascontiguousarray(g[f"{tag}_tus"]).view(synth.HEVC_TU_DTYPE).reshape(-1).copy()
        rc, tk, wt, st = plan(tus, w, h, w // 2, h // 2, 6)
        assert rc == 0 and int(st[1]) == 6, (tag, list(st))
        wins = {(int(t["cidx"]), int(t["y"]) >> (6 if t["cidx"] == 0 else 5), int(t["x"]) >> (6 if t["cidx"] == 0 else 5)) for t in tus}
        assert int(st[0]) == len(wins), (tag, int(st[0]), len(wins))
