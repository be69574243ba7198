#!/bin/bash
# Round-6 evidence run on the MI355X box: rocprofv3 kernel statistics and PMC passes for bench.py, the f1 row and the stage benches.
# usage (from the repo root on the box): bash tests/tools/collect_profiles_r6.sh [main]   -> gpurun_out/profiles_r6/ (main: without the subsequence decoder's own collection, tests/tools/collect_profiles_r5_huff.sh)   (copy what is kept into profiles/r6_*)
# (every step prints a line when it is done: the box's watchdog kills a command that is silent for seven minutes)
set -u
R=$PWD
O=$R/gpurun_out/profiles_r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # name, program args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name --output-format csv -- python3 "$@" > $O/$name.stdout 2> $O/$name.stderr
  find /tmp/rp_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  echo "done $name"
}
pmc() { # name, kernel filter, counters..., then "--", program args
  local name=$1 kern=$2; shift 2
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --pmc "${ctrs[@]}" -d /tmp/rp_$name -o pmc --output-format csv -- python3 "$@" > /dev/null 2>&1
  python3 $R/tests/tools/pmc_summary.py /tmp/rp_$name --kernel $kern >> $O/$name.txt
  echo "done $name"
}
# the driver's command, with the verbose record
prof bench $R/bench.py --extra-file $O/bench_full.json
grep '^{' $O/bench.stdout | tail -1 > $O/bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_pmc_$c
  rocprofv3 --pmc $c -d /tmp/rp_pmc_$c -o pmc --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > /dev/null 2>&1
  python3 $R/tests/tools/pmc_summary.py /tmp/rp_pmc_$c --kernel k_jpeg420 > $O/pmc_$c.txt
  echo "done pmc $c"
done
# f1: the device Huffman stage (256 x 4K files with one restart interval per MCU row) -- kernel statistics and one PMC pass
# the JPEG entropy front end: its own collection (the subsequence decoder; FFHIP_JPEG_SYNC=0 below for round 4's lane-per-interval kernel)
FFHIP_JPEG_SYNC=0 F1_TAGS=dri_per_mcu_row prof huff_gpu $R/bench.py --steps 2 --warmup 1 --no-cpu --extras f1
FFHIP_JPEG_SYNC=0 F1_TAGS=dri_per_mcu_row pmc huff_gpu_pmc k_jpeg_huff SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES -- $R/bench.py --steps 2 --warmup 1 --no-cpu --extras f1
FFHIP_JPEG_SYNC=0 F1_TAGS=dri_per_mcu_row pmc huff_gpu_pmc k_jpeg_huff SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM -- $R/bench.py --steps 2 --warmup 1 --no-cpu --extras f1
# the single 8K picture and the grids
prof intra_c5 $R/tests/tools/bench_intra_c5.py 6
pmc intra_c5_pmc k_hevc_intra_groups SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $R/tests/tools/bench_intra_c5.py 6
PICTURES=8 NO_CPU=1 prof hevc_grid8 $R/tests/tools/bench_hevc_grid.py
PICTURES=8 NO_CPU=1 pmc hevc_grid8_pmc k_hevc_intra_groups SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- $R/tests/tools/bench_hevc_grid.py
cd $R
NO_CPU=1 python3 tests/tools/bench_hevc_grid.py > $O/hevc_grid.json 2> /dev/null
echo "done hevc grid"
# VP8 frames: the kernel at 256 frames, the sweep with the waves-per-frame choice
export MODE=frames FFHIP_VP8_FRAMES=fused FRAMES=256
cd /tmp
prof vp8_frames256 $R/tests/tools/prof_vp8_batch.py
unset MODE FFHIP_VP8_FRAMES FRAMES
cd $R
SIZES=16,64,256,1024 python3 tests/tools/bench_vp8_frames.py > $O/vp8_frames.jsonl 2> /dev/null
echo "done vp8 frames"
# timelines: the plain call and the pipelined tile call (its pre-pass next to the colour conversion and the next residual batches)
bash tests/tools/prof_hevc_timeline.sh > /dev/null 2>&1; cp gpurun_out/hevc_timeline/grid_8.txt $O/hevc_timeline_grid8.txt; cp gpurun_out/hevc_timeline/grid_1.txt $O/hevc_timeline_grid1.txt; cp gpurun_out/hevc_timeline/one_8k_c5.txt $O/hevc_timeline_8k_c5.txt
echo "done timelines"
ls -la $O | tail -40
# the headline ALONE (bench.py --no-extra: the same K launches, no other workload of the same kernel name in the statistics)
cd /tmp
prof bench_headline $R/bench.py --no-extra
grep '^{' $O/bench_headline.stdout | tail -1 > $O/bench_headline.json
cd $R
cd $R
if [ "${1:-all}" != "main" ]; then bash tests/tools/collect_profiles_r5_huff.sh > $O/huff_sync.log 2>&1; echo "done subsequence decoder"; fi
# the store-shape / placement microbenchmark: several allocations of the output buffer held at once
cd $R
if [ -x tests/tools/membench_jpeg_rows.bin ]; then timeout -k 10 200 tests/tools/membench_jpeg_rows.bin 7 0 > $O/membench_rows.txt 2>&1; echo "done membench rows"; fi
