#!/usr/bin/env python3
"""Per-configuration kernel time of k_hevc_intra_groups from a rocprofv3 --kernel-trace run of diag_intra_latency.py:
   cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- python3 REPO/tests/tools/diag_intra_latency.py
   python3 diag_intra_kernel_times.py OUT
diag_intra_latency.py launches the kernel 23 times per configuration, in a fixed order of 24 configurations."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_hevc_intra_groups" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
names = [(n, m) for n in (4, 8, 16, 32) for m in ("DC+res", "planar+filter", "ang34+filter", "hor+rdpcm", "ang20", "DC nores")]
tus = {4: 256, 8: 64, 16: 16, 32: 4}
per = len(rows) // len(names)
for i, (n, m) in enumerate(names):
    d = [(e - s) / 1e3 for s, e in rows[i * per:(i + 1) * per]][3:]
    avg = sum(d) / len(d)
    print(f"n={n:2d} {m:14s} kernel {avg:7.1f} us  -> {avg / tus[n]:5.2f} us/TU")
