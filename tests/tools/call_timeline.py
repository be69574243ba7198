#!/usr/bin/env python3
"""The LAST call of a traced program as a timeline of its kernels AND its memory copies (rocprofv3 --kernel-trace --memory-copy-trace, CSV):
start offset, duration, what.  usage: call_timeline.py <dir> <kernel that ends the call (substring)> [gap_us=2500] [min_us=20]
The call is what lies behind the last gap of more than gap_us without device activity in front of the last kernel matching."""
import csv, glob, os, sys
d, last = sys.argv[1], sys.argv[2]
gap = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 2.5e6
min_ns = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 2e4
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel  " + r["Kernel_Name"][:60]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy    " + r["Direction"].replace("MEMORY_COPY_", "")))
rows.sort()
ends = [i for i, r in enumerate(rows) if last in r[2]]
if not ends: sys.exit("no kernel matching " + last)
e = ends[-1]
s = e
busy_to = rows[e][0]
while s > 0 and busy_to - max(r[1] for r in rows[max(0, s - 8):s]) < gap:
    s -= 1
    busy_to = rows[s][0]
t0 = rows[s][0]
shown = 0
for st, en, name in rows[s:e + 1]:
    if en - st >= min_ns:
        print("%9.1f us  +%8.1f us  %s" % ((st - t0) / 1e3, (en - st) / 1e3, name))
    else:
        shown += 1
print("span %.1f us; %d shorter than %.0f us not shown" % ((max(r[1] for r in rows[s:e + 1]) - t0) / 1e3, shown, min_ns / 1e3))
