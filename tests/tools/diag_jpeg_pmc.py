#!/usr/bin/env python3
"""Placement modes of the headline kernel under rocprofv3 --pmc: the output buffer is re-allocated TRIALS times, each
placement gets 4 launches (the last one timed with HIP events and printed as `T <trial> <ms>`).  Run as
  rocprofv3 --pmc <counters> -d DIR -o pmc --output-format csv -- python3 tests/tools/diag_jpeg_pmc.py > times.txt
then `python3 tests/tools/diag_jpeg_pmc.py --join DIR times.txt` prints counters of every placement's last launch."""
import os, sys, json
if "--join" in sys.argv:
    import csv, glob
    from collections import defaultdict
    d, tf = sys.argv[sys.argv.index("--join") + 1:][:2]
    times = [float(l.split()[2]) for l in open(tf) if l.startswith("T ")]
    per = defaultdict(lambda: defaultdict(float))
    inst = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_jpeg420" in row["Kernel_Name"]:
                per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
                inst[int(row["Dispatch_Id"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    ids = sorted(per)
    names = sorted({c for v in per.values() for c in v})
    print("trial ms " + " ".join(names))
    for t in range(len(times)):
        k = ids[4 * t + 3] if 4 * t + 3 < len(ids) else None
        if k is None: break
        print(t, times[t], " ".join(f"{per[k][c]:.0f}" + (f"(x{len(inst[k][c])} max/mean {max(inst[k][c]) * len(inst[k][c]) / max(per[k][c], 1):.2f})" if len(inst[k][c]) > 1 else "") for c in names))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffpic_amd import capi, ops, synth
dev = torch.device("cuda", 0)
L = capi.require_device(0)
st = torch.cuda.current_stream().cuda_stream
e0, e1 = L.ffhip_event_create(), L.ffhip_event_create()
n = 256
cols, rows = 240, 135
geom = capi.jpeg_geom(cols, rows)
H, W = geom.height, geom.width
mcus = cols * rows
q = torch.from_numpy(synth.quant_tables().astype(np.int16)).to(dev)
ty = torch.randint(-30, 31, (n * mcus * 4, 64), device=dev, dtype=torch.int16)
tu = torch.randint(-30, 31, (n * mcus, 64), device=dev, dtype=torch.int16)
tv = torch.randint(-30, 31, (n * mcus, 64), device=dev, dtype=torch.int16)
for trial in range(int(os.environ.get("TRIALS", "10"))):
    dummy = torch.empty(1 + (trial % 7) * 37_000_003, dtype=torch.uint8, device=dev)
    out = torch.empty(n * W * 4 * H, dtype=torch.uint8, device=dev)
    del dummy
    def step():
        ops.jpeg_recon_batch(geom, n, ty.data_ptr(), tu.data_ptr(), tv.data_ptr(), q.data_ptr(), 0, out.data_ptr(), W * 4, W * 4 * H, None, 0, st)
    for _ in range(3): step()
    L.ffhip_event_record(e0, st); step(); L.ffhip_event_record(e1, st)
    print("T", trial, round(L.ffhip_event_elapsed_ms(e0, e1), 4), hex(out.data_ptr()), flush=True)
    del out
    torch.cuda.empty_cache()
