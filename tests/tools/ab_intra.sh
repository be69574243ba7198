#!/bin/bash
# A/B of two builds of the library (ffpic_amd/libffpic_hip_A.so, _B.so) in ONE gpurun call (boxes differ by several %):
# kernel-only per-TU times of the single-group lists and the 8K pictures, each variant twice, interleaved.
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in A B; do
  export FFHIP_LIB=libffpic_hip_$v.so
  rm -rf /tmp/lat_$v; rocprofv3 --kernel-trace -d /tmp/lat_$v -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/tools/diag_intra_latency.py > /dev/null 2>&1
  echo "== $v (rep $rep)"; python3 $GRAFT_REPO_ROOT/tests/tools/diag_intra_kernel_times.py /tmp/lat_$v | awk '{printf "%s %s %s | ", $2, $3, $(NF-1)} END {print ""}'
  python3 $GRAFT_REPO_ROOT/tests/tools/bench_intra_c5.py 6 2>&1 | tail -1
done; done
