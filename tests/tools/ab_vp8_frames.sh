#!/bin/bash
# A/B of builds of the library (ffpic_amd/libffpic_hip_<V>.so, tests/tools/build_variant.sh) in ONE gpurun call: the fused VP8 frame kernel
# at 256 and 1024 frames, each variant twice, interleaved.  VARIANTS="A B C"  WAVES=8  SIZES=256,1024
for rep in 1 2; do for v in ${VARIANTS:-A B}; do
  echo "== $v (rep $rep)"
  FFHIP_LIB=libffpic_hip_$v.so SIZES=${SIZES:-256,1024} SOURCES=${SOURCES:-encoder} WAVES=${WAVES:-8} timeout -k 10 240 python3 tests/tools/bench_vp8_frames.py 2>&1 | tail -4
done; done
