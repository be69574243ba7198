# SQ counters of k_vp8_frames at FRAMES frames (default 1024): waits, vector-memory and LDS levels.  (TCP_* latency counters abort rocprofv3 on this
# image and leave the run hanging until the watchdog: not asked for.)
set -u
R=$PWD; O=$R/gpurun_out/r4_vp8_frames_lat; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MODE=frames FFHIP_VP8_FRAMES=fused FRAMES=${FRAMES:-1024}
pm() { name=$1; shift; rm -rf /tmp/rp_$name; rocprofv3 --pmc "$@" -d /tmp/rp_$name -o pmc --output-format csv -- python3 $R/tests/tools/prof_vp8_batch.py > $O/$name.out 2>&1; python3 $R/tests/tools/pmc_summary.py /tmp/rp_$name --kernel k_vp8_frames > $O/$name.txt 2>&1; echo done $name; }
pm a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pm b SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU
cat $O/a.txt $O/b.txt
