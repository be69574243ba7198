#!/bin/bash
# A/B of two builds on the headline bench line through the same launcher: ffpic_amd/libffpic_hip_A.so and _B.so, alternating, in ONE gpurun call.
set -u
O=gpurun_out/ab_bench.txt
: > $O
run() { python3 -c "import sys, os, runpy; sys.argv=['bench.py','--no-cpu','--no-extra','--steps','30']; import ffpic_amd.capi as c; c.LIB_PATH=os.path.join(os.path.dirname(c.LIB_PATH),'libffpic_hip_$1.so'); runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline']['copy_kernel_GBps'])" >> $O; }
for rep in 1 2 3; do for l in ${@:-A B}; do run $l; done; done
cat $O
