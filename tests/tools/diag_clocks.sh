#!/bin/bash
# samples rocm-smi clocks / power while the variance diagnostic runs (one GPU process + rocm-smi readers)
set -u
O=gpurun_out/clocks.txt
: > $O
TRIALS=${TRIALS:-10} python3 tests/tools/diag_jpeg_variance.py > gpurun_out/jpeg_variance2.txt 2>&1 &
PID=$!
for i in $(seq 1 40); do
  if ! kill -0 $PID 2>/dev/null; then break; fi
  echo "--- sample $i $(date +%s.%N)" >> $O
  rocm-smi -c -P -t 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|power\|Temperature (Sensor junction)\|memory)" >> $O
  sleep 0.4
done
wait $PID
cat gpurun_out/jpeg_variance2.txt | cut -c1-110
tail -60 $O
