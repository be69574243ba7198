R=$PWD; cd /tmp; export TMPDIR=/tmp
for l in libffpic_hip_exp4.so libffpic_hip_expnodc.so; do cd $R; FFHIP_LIB=$l FRAMES=256 timeout -k 10 300 python tests/tools/diag_vp8_batch_waves.py 2>/dev/null | tail -1 | cut -c1-150; done
cd /tmp
rm -rf /tmp/rp_x; MODE=pred rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d /tmp/rp_x -o pmc --output-format csv -- python3 $R/tests/tools/prof_vp8_batch.py > /dev/null 2>&1; python3 $R/tests/tools/pmc_summary.py /tmp/rp_x --kernel k_vp8_predict
rm -rf /tmp/rp_y; MODE=pred rocprofv3 --pmc SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY -d /tmp/rp_y -o pmc --output-format csv -- python3 $R/tests/tools/prof_vp8_batch.py > /dev/null 2>&1; python3 $R/tests/tools/pmc_summary.py /tmp/rp_y --kernel k_vp8_predict
