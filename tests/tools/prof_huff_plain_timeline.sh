#!/bin/bash
# The kernels of ONE ffhip_jpeg_entropy_batch_gpu call on files without restart markers (rounds, scan, write pass, DC sums) as a timeline.
# -> gpurun_out/huff_plain/timeline.txt
set -u
R=$PWD
O=$R/gpurun_out/huff_plain
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hp
rocprofv3 --kernel-trace -d /tmp/rp_hp -o hp --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > $O/run.json 2> $O/err.txt
python3 $R/tests/tools/kernel_timeline.py /tmp/rp_hp "k_huff_span<0" k_huff_sync_verdict 2 > $O/timeline.txt
cat $O/run.json
