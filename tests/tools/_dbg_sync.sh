set -u
R=$PWD
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3 4; do
rm -rf /tmp/rp_hp
FFHIP_DBG_SYNC=$d FFHIP_JPEG_SYNC_PARTS=1 timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/rp_hp -o hp --output-format csv -- python3 $R/tests/tools/bench_huff_plain.py > /dev/null 2>&1
echo "dbg $d: $(python3 $R/tests/tools/kernel_timeline.py /tmp/rp_hp "k_huff_sync_segs" k_huff_sync_verdict 1 | grep "span<2")"
done
