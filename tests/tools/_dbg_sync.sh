set -u
R=$PWD
python3 -m pytest tests/test_huff_gpu.py tests/test_switches_gpu.py -x -q -m gpu -k "huff or plain or few_rounds" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for so in 1 0; do
rm -rf /tmp/rp_hp
FFHIP_JPEG_SYNC_SORT=$so FFHIP_JPEG_SYNC_PARTS=1 rocprofv3 --kernel-trace -d /tmp/rp_hp -o hp --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > /dev/null 2>&1
echo "sort $so"; python3 $R/tests/tools/kernel_timeline.py /tmp/rp_hp "k_huff_span<0" k_huff_sync_verdict 2 | grep -v "+      [0-9]\.[0-9] us"
done
cd $R
echo "sorted:   $(STREAM=1 python3 tests/tools/bench_huff_plain.py 2>&1 | tail -1)"
echo "unsorted: $(FFHIP_JPEG_SYNC_SORT=0 STREAM=1 python3 tests/tools/bench_huff_plain.py 2>&1 | tail -1)"
