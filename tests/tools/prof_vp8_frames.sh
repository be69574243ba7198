# rocprofv3 on ffhip_vp8_decode_frames' fused kernel, FRAMES copies of the encoder's 1080p frame (default 256): SQ counters in two passes, then kernel stats
set -u
R=$PWD; O=$R/gpurun_out/r4_vp8_frames_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MODE=frames FFHIP_VP8_FRAMES=fused
pm() { name=$1; shift; rm -rf /tmp/rp_$name; rocprofv3 --pmc "$@" -d /tmp/rp_$name -o pmc --output-format csv -- python3 $R/tests/tools/prof_vp8_batch.py > $O/$name.out 2>&1; python3 $R/tests/tools/pmc_summary.py /tmp/rp_$name --kernel k_vp8_frames > $O/$name.txt 2>&1; echo done $name; }
pm a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pm b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pm c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS
rm -rf /tmp/rp_ks; rocprofv3 --kernel-trace --stats -d /tmp/rp_ks -o ks --output-format csv -- python3 $R/tests/tools/prof_vp8_batch.py > $O/ks.out 2>&1; find /tmp/rp_ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cat $O/a.txt $O/b.txt $O/c.txt; head -8 $O/kernel_stats.csv
