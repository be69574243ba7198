#!/bin/bash
# The kernels of ONE ffhip_hevc_intra_recon_tiles call (the tile loop as a pipeline of chunks) as a timeline, for the tile grid at PICTURES.
# CHUNKS = number of chunks the call is cut into (FFHIP_HEVC_TILE_CHUNKS).  -> gpurun_out/hevc_timeline/tiles_<pictures>_<chunks>.txt
set -u
R=$PWD
O=$R/gpurun_out/hevc_timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in ${PICTURES_LIST:-8 1}; do
  for c in ${CHUNKS_LIST:-4}; do
    rm -rf /tmp/rp_tl
    FFHIP_HEVC_TILE_CHUNKS=$c PICTURES=$p NO_CPU=1 rocprofv3 --kernel-trace -d /tmp/rp_tl -o tl --output-format csv -- python3 $R/tests/tools/bench_hevc_grid.py > $O/tiles_${p}_${c}.json 2> $O/err_tiles_$p.txt
    python3 $R/tests/tools/kernel_timeline.py /tmp/rp_tl k_plan_init k_hevc_intra_serial $c > $O/tiles_${p}_${c}.txt
    echo "done tiles $p x $c"
  done
done
