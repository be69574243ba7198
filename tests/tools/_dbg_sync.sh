set -u
R=$PWD
python3 -m pytest tests/test_huff_gpu.py -x -q -m gpu -k "plain or generated" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hp
FFHIP_JPEG_SYNC_PARTS=1 rocprofv3 --kernel-trace -d /tmp/rp_hp -o hp --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > /dev/null 2>&1
python3 $R/tests/tools/kernel_timeline.py /tmp/rp_hp "k_huff_span<0" k_huff_sync_verdict 2 | grep "span<2\|span<0"
