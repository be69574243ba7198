#!/bin/bash
# subsequence sizes for the files without restart markers: one timeline per size -> gpurun_out/huff_plain/timeline_<bits>.txt
set -u
R=$PWD
O=$R/gpurun_out/huff_plain
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in ${BITS_LIST:-512 1024 2048 4096 8192}; do
  rm -rf /tmp/rp_hp
  FFHIP_JPEG_SYNC_BITS=$b rocprofv3 --kernel-trace -d /tmp/rp_hp -o hp --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > $O/run_$b.json 2> $O/err_$b.txt
  python3 $R/tests/tools/kernel_timeline.py /tmp/rp_hp "k_huff_span<0" k_huff_sync_verdict 2 > $O/timeline_$b.txt
  echo "bits $b: $(cat $O/run_$b.json)"; grep -v "sync_list\|  +      [0-9]\.[0-9] us" $O/timeline_$b.txt
done
