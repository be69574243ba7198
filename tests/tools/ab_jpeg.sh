#!/bin/bash
# A/B of two builds (ffpic_amd/libffpic_hip_A.so against the default build) on the JPEG kernels, in ONE gpurun call:
# the stage bench at 64 and 256 images and the headline bench, alternating so that box drift shows up.
set -u
O=gpurun_out/ab_jpeg.txt
: > $O
benchA() { python3 -c "import sys, os, runpy; sys.argv=['bench.py','--no-cpu','--no-extra','--steps','30']; import ffpic_amd.capi as c; c.LIB_PATH=os.path.join(os.path.dirname(c.LIB_PATH),'libffpic_hip_A.so'); runpy.run_path('bench.py', run_name='__main__')" ; }
for rep in 1 2; do
  for n in 64 256; do
    echo "A images $n" >> $O; FFHIP_LIB=libffpic_hip_A.so FFHIP_BENCH_IMAGES=$n python3 tests/tools/bench_jpeg_geoms.py 2>/dev/null | tr -d "\n " >> $O; echo >> $O
    echo "B images $n" >> $O; FFHIP_BENCH_IMAGES=$n python3 tests/tools/bench_jpeg_geoms.py 2>/dev/null | tr -d "\n " >> $O; echo >> $O
  done
  echo "A bench" >> $O; benchA 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" >> $O
  echo "B bench" >> $O; python3 bench.py --no-cpu --no-extra --steps 30 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'])" >> $O
done
cat $O
