# rocprofv3 SQ counters of KERNEL (default k_hevc_intra_groups) on the PICTURES-picture tile grid (default 8) -> gpurun_out/r4_hevc_grid_pmc/
set -u
R=$PWD; O=$R/gpurun_out/r4_hevc_grid_pmc; mkdir -p $O
export PICTURES=${PICTURES:-8} NO_CPU=1
KERNEL=${KERNEL:-k_hevc_intra_groups}
cd /tmp && export TMPDIR=/tmp
pm() { name=$1; shift; rm -rf /tmp/rp_$name; rocprofv3 --pmc "$@" -d /tmp/rp_$name -o pmc --output-format csv -- python3 $R/tests/tools/bench_hevc_grid.py > $O/$name.out 2>&1; python3 $R/tests/tools/pmc_summary.py /tmp/rp_$name --kernel $KERNEL > $O/$name.txt 2>&1; echo done $name; }
pm a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pm b SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
[ -n "${PASSES_AB:-}" ] || pm c SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM
cat $O/a.txt $O/b.txt $O/c.txt
