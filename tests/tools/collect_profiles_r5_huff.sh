#!/bin/bash
# Round-5 evidence for the subsequence decoder (ffhip_huff_gpu.hip, k_huff_span): kernels of one call on 256 4K files with and without restart markers, the same
# call as the pipeline of parts it is (copies included), kernel statistics and PMC passes.   -> gpurun_out/profiles_r5_huff/   (kept: profiles/r5_huff_sync_*)
set -u
R=$PWD
O=$R/gpurun_out/profiles_r5_huff
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for kind in plain dri; do
  [ $kind = dri ] && export RESTART_ROWS=1 || unset RESTART_ROWS
  rm -rf /tmp/rp_hs
  FFHIP_JPEG_SYNC_PARTS=1 rocprofv3 --kernel-trace --stats -d /tmp/rp_hs -o hs --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > $O/${kind}_one_part.json 2> /dev/null
  python3 $R/tests/tools/kernel_timeline.py /tmp/rp_hs "k_huff_span<0" k_huff_sync_verdict 2 > $O/${kind}_one_part_kernels.txt
  find /tmp/rp_hs -name "*kernel_stats.csv" -exec cp {} $O/${kind}_one_part_kernel_stats.csv \;
  rm -rf /tmp/rp_hs
  STREAM=1 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/rp_hs -o hs --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > $O/${kind}_parts_traced.json 2> /dev/null
  python3 $R/tests/tools/call_timeline.py /tmp/rp_hs k_huff_sync_verdict > $O/${kind}_parts_timeline.txt
  STREAM=1 python3 $R/tests/tools/bench_huff_plain.py > $O/${kind}_parts.json 2> /dev/null
  echo "done $kind: $(cat $O/${kind}_parts.json)"
done
unset RESTART_ROWS
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/rp_hs
  FFHIP_JPEG_SYNC_PARTS=1 rocprofv3 --pmc $set -d /tmp/rp_hs -o pmc --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > /dev/null 2>&1
  for k in "k_huff_span<0" "k_huff_span<1" "k_huff_span<2"; do
    echo "== $k" >> $O/pmc.txt
    python3 $R/tests/tools/pmc_summary.py /tmp/rp_hs --kernel "$k" >> $O/pmc.txt
  done
  echo "done pmc set"
done
cd $R
python3 bench.py --extras f1 --extra-file $O/bench_f1_full.json > $O/bench_f1.json 2> /dev/null
echo "done bench f1"
ls -la $O
