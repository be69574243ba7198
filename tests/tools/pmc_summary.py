#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch.
usage: pmc_summary.py <dir-or-csv>... [--kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kern = None
    if "--kernel" in sys.argv:
        kern = sys.argv[sys.argv.index("--kernel") + 1]
        args = [a for a in args if a != kern]
    files = []
    for a in args:
        files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if kern and kern not in name:
                continue
            acc[name][row["Counter_Name"]].append((row["Dispatch_Id"], float(row["Counter_Value"])))
    for name, ctrs in acc.items():
        print(name[:80])
        for c, vals in sorted(ctrs.items()):
            per = defaultdict(float)
            for d, v in vals:
                per[d] += v
            xs = list(per.values())
            print(f"  {c:28s} mean/dispatch {sum(xs)/len(xs):18.1f}   n={len(xs)}")

main()
