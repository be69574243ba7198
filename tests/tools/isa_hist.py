#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing: isa_hist.py file.s mangled-name-substring [--body]."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M)
body = m.group(2)
c = Counter()
for l in body.splitlines():
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    c[l.split()[0]] += 1
kinds = Counter()
for k, v in c.items():
    kinds['VALU' if k.startswith('v_') else 'SALU' if k.startswith('s_') else 'LDS' if k.startswith('ds_') else 'VMEM'] += v
print(m.group(1), dict(kinds))
if '--body' in sys.argv:
    print(body)
else:
    for k, v in c.most_common(70):
        print(f"{v:5d} {k}")
