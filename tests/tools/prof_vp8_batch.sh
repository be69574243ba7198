set -u
R=$PWD; O=$R/gpurun_out/r3_vp8_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
pm() { name=$1; shift; rm -rf /tmp/rp_$name; rocprofv3 --pmc "$@" -d /tmp/rp_$name -o pmc --output-format csv -- python3 $R/tests/tools/prof_vp8_batch.py > $O/$name.out 2>&1; python3 $R/tests/tools/pmc_summary.py /tmp/rp_$name --kernel k_vp8 > $O/$name.txt 2>&1; echo done $name; }
pm a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pm b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
rm -rf /tmp/rp_ks; rocprofv3 --kernel-trace --stats -d /tmp/rp_ks -o ks --output-format csv -- python3 $R/tests/tools/prof_vp8_batch.py > $O/ks.out 2>&1; find /tmp/rp_ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cat $O/a.txt $O/b.txt; head -8 $O/kernel_stats.csv
