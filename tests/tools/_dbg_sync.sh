set -u
R=$PWD
python3 -m pytest tests/test_huff_gpu.py tests/test_switches_gpu.py -x -q -m gpu -k "huff or plain or few_rounds" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hp
FFHIP_JPEG_SYNC_PARTS=1 timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/rp_hp -o hp --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > /dev/null 2>&1
python3 $R/tests/tools/kernel_timeline.py /tmp/rp_hp "k_huff_sync_segs" k_huff_sync_verdict 1 | grep -v "+      [0-9]\.[0-9] us"
cd $R
for i in 1 2; do echo "parts: $(STREAM=1 python3 tests/tools/bench_huff_plain.py 2>&1 | tail -1)"; done
