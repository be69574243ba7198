#!/bin/bash
# placement modes of the headline kernel against memory-side counters (separate --pmc passes; the modes are per process, so each pass prints its own times)
set -u
R=$PWD; O=$R/gpurun_out/jpeg_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in ${SETS:-"TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUSY_sum TCC_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ TCC_EA0_WRREQ"}; do
  i=$((i+1)); rm -rf /tmp/jp_$i
  rocprofv3 --pmc $set -d /tmp/jp_$i -o pmc --output-format csv -- python3 $R/tests/tools/diag_jpeg_pmc.py > $O/times_$i.txt 2> $O/err_$i.txt
  python3 $R/tests/tools/diag_jpeg_pmc.py --join /tmp/jp_$i $O/times_$i.txt > $O/join_$i.txt 2>&1
  echo "== pass $i"; cat $O/join_$i.txt
done
