#!/bin/bash
# rocprofv3 kernel statistics of ONE grid size of tests/tools/bench_hevc_grid.py (PICTURES, default 8): which kernels an
# 1.8-million-TU ffhip_hevc_intra_recon call consists of.  -> gpurun_out/r3_hevc_grid_prof/
set -u
R=$PWD
O=$R/gpurun_out/r3_hevc_grid_prof
mkdir -p $O
export PICTURES=${PICTURES:-8} NO_CPU=1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_grid
rocprofv3 --kernel-trace --stats -d /tmp/rp_grid -o grid --output-format csv -- python3 $R/tests/tools/bench_hevc_grid.py > $O/grid.stdout 2> $O/grid.stderr
find /tmp/rp_grid -name "*kernel_stats.csv" -exec cp {} $O/grid_kernel_stats.csv \;
cut -d, -f1-4 $O/grid_kernel_stats.csv | cut -c1-120 | head -30
