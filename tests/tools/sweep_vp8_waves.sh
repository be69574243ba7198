for cfg in "0 0" "600 256" "800 256" "400 384" "600 384" "800 512" "1200 512"; do set -- $cfg
  if [ $1 = 0 ]; then unset FFHIP_VP8_PRED_WAVES FFHIP_VP8_LF_WAVES; else export FFHIP_VP8_PRED_WAVES=$1 FFHIP_VP8_LF_WAVES=$2; fi
  echo "cfg $cfg"; SOURCES=encoder SIZES=16,64 timeout -k 10 200 python3 tests/tools/bench_vp8_frames.py 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('  ',r['source'],r['frames'],r['rows_ms'])
" || exit 1
done
