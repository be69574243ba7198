#!/usr/bin/env python3
"""The LAST occurrence of a chain of kernels in a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue.
usage: kernel_timeline.py <dir with *_kernel_trace.csv> <first kernel of the chain (substring)> <last kernel (substring)> [n]
n (default 1): the chain starts at the n-th last kernel matching <first> in front of the end (a pipelined call has one per chunk)"""
import csv, glob, os, sys
d, first, last = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if last in r[2]]
if not ends: sys.exit("no kernel matching " + last)
e = ends[-1]
back = int(sys.argv[4]) if len(sys.argv) > 4 else 1
s = sorted(i for i in range(e + 1) if first in rows[i][2])[-back]
t0 = rows[s][0]
tot = 0
for st, en, name, q in rows[s:e + 1]:
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((st - t0) / 1e3, (en - st) / 1e3, q, name[:70]))
    tot += en - st
print("span %.1f us, sum of kernels %.1f us" % ((max(r[1] for r in rows[s:e + 1]) - t0) / 1e3, tot / 1e3))
