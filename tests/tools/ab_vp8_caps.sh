#!/bin/bash
# The VP8 chain at small batches against the wave caps (FFHIP_VP8_PRED_WAVES / FFHIP_VP8_LF_WAVES): default caps, then explicit ones.
# SIZES (default 16,64).  One gpurun call so that the box is the same.
export SIZES=${SIZES:-16,64}
show() { python3 -c "import sys,json; d=json.load(sys.stdin); print(sys.argv[1], {k:[(r['frames'], r['chain_ms']) for r in d[k]] for k in ('encoder','random')})" "$1"; }
for rep in 1 2; do
  python3 tests/tools/bench_vp8_batch_sweep.py 2>/dev/null | show default
  for pw in ${CAPS:-256:256 384:256 512:256 384:128 384:384}; do
    FFHIP_VP8_PRED_WAVES=${pw%%:*} FFHIP_VP8_LF_WAVES=${pw##*:} python3 tests/tools/bench_vp8_batch_sweep.py 2>/dev/null | show "pred:lf=$pw"
  done
done
