"""Build-time guard for the dependency-scheduled kernels (no GPU): every ticket a wave takes -- `if (lane == 0) t = atomic_fetch_add(counter, 1);
t = readfirstlane(t); if (t >= n) leave;` inside a loop -- must compile to ONE loop that holds the atomic, the v_readfirstlane and the compare.

Round 3 met a form of k_vp8_predict_rows that never terminated (profiles/r5_vp8_hang_isa.txt has the ISA; tests/tools/experiments/
r3_vp8_ticket_loop_per_form.patch the source): with a ticket loop per form, each left by `return`, the compiler rotated the loop and the
structuriser put the `lane == 0` test into the latch of an INNER loop -- lane 0 left it to take the next ticket, lanes 1..63 stayed, set their
own `ticket = 0` and went round again: v_readfirstlane then read lane 1's 0, no atomic was executed on that path, and the row of ticket 0 was
decoded for ever.  Per lane that is the same program; across lanes it is not, because readfirstlane is a convergent operation.  The source
cannot rule it out, so the compiled code is checked: between a ticket's atomic and the compare that consumes it no new loop may begin."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ffpic_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FILES = ["ffhip_vp8_pred.hip", "ffhip_vp8_lf.hip", "ffhip_hevc_intra.hip"]


def ticket_sites(asm):
    """(kernel, line, ok, window) for every returning global atomic add: ok = walking on from it (through unconditional branches), the first
    unsigned compare comes before any new loop header"""
    out = []
    kernel = None
    lines = asm.split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel = m.group(1)
        if "global_atomic_add" in l and " sc0" in l and not l.strip().startswith(";"):
            ok, window, k = None, [], i + 1
            for _ in range(400):
                if k >= len(lines):
                    break
                l2 = lines[k]
                window.append(l2)
                if "Loop Header" in l2:
                    ok = False
                    break
                if re.search(r"\b[sv]_cmp_(ge|lt|gt|le)_u32(_e32|_e64)?\b", l2):   # (the uniform compare may come as a vector compare into vcc)
                    ok = True
                    break
                mb = re.match(r"^\s*s_branch\s+(\.LBB\d+_\d+)", l2)
                k = labels[mb.group(1)] if mb and mb.group(1) in labels else k + 1
            out.append((kernel, i + 1, ok, window))
    return out


def test_the_lint_sees_the_round_3_hang():
    """the reconstructed ISA of the hung arrangement (an excerpt kept under profiles/): the lint must refuse it, and accept the shipped form next to it"""
    txt = open(os.path.join(ROOT, "profiles", "r5_vp8_hang_isa.txt")).read()
    bad, good = txt.split("The shipped arrangement")
    # the excerpt leaves the row's body out ("..."): the atomic, then the inner loop's header in front of the compare
    assert [ok for _, _, ok, _ in ticket_sites(bad)] == [False]
    assert [ok for _, _, ok, _ in ticket_sites(good)] == [True]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (cross-compiles without a GPU)")
def test_every_ticket_loop_keeps_atomic_and_compare_in_one_loop(tmp_path):
    procs = []
    for f in FILES:
        out = str(tmp_path / (f + ".s"))
        procs.append((f, out, subprocess.Popen([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form",
                                                "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, f)],
                                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    seen = 0
    for f, out, p in procs:
        assert p.wait(timeout=600) == 0, f
        sites = ticket_sites(open(out).read())
        assert sites, f"{f}: no ticket atomics found -- the lint no longer recognises them"
        for kernel, line, ok, window in sites:
            assert ok, f"{f}:{line} ({kernel}): a loop begins between a ticket's atomic and the compare that consumes it:\n" + "\n".join(window[-12:])
            seen += 1
    assert seen >= 8      # prediction (x2 instances), loop filter (x2), HEVC grouped kernel (x2 instances, sharded + single counter), ...
