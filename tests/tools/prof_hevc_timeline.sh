#!/bin/bash
# The kernels of ONE ffhip_hevc_intra_recon call as a timeline (start, duration, queue), for the tile grid at PICTURES (default "8") and for
# one 8K picture.  -> gpurun_out/hevc_timeline/
set -u
R=$PWD
O=$R/gpurun_out/hevc_timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in ${PICTURES_LIST:-8 1}; do
  rm -rf /tmp/rp_tl
  PICTURES=$p NO_CPU=1 rocprofv3 --kernel-trace -d /tmp/rp_tl -o tl --output-format csv -- python3 $R/tests/tools/bench_hevc_grid.py > /dev/null 2> $O/err_$p.txt
  python3 $R/tests/tools/kernel_timeline.py /tmp/rp_tl k_plan_init k_hevc_intra_serial > $O/grid_$p.txt
  echo "done grid $p"
done
rm -rf /tmp/rp_tl
rocprofv3 --kernel-trace -d /tmp/rp_tl -o tl --output-format csv -- python3 $R/tests/tools/bench_intra_c5.py 6 > /dev/null 2> $O/err_c5.txt
python3 $R/tests/tools/kernel_timeline.py /tmp/rp_tl k_plan_init k_hevc_intra_serial > $O/one_8k.txt
rm -rf /tmp/rp_tl
MIXES=c5mix rocprofv3 --kernel-trace -d /tmp/rp_tl -o tl --output-format csv -- python3 $R/tests/tools/bench_intra_c5.py 6 > /dev/null 2>> $O/err_c5.txt
python3 $R/tests/tools/kernel_timeline.py /tmp/rp_tl k_plan_init k_hevc_intra_serial > $O/one_8k_c5.txt
echo "done 8k"
