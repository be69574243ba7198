#!/bin/bash
# Round-3 evidence run on the MI355X box: rocprofv3 kernel statistics and PMC passes for bench.py and the stage benches.
# usage (from the repo root on the box): bash tests/tools/collect_profiles.sh   -> gpurun_out/profiles_r3/
set -u
R=$PWD
O=$R/gpurun_out/profiles_r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # name, program args...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --kernel-trace --stats -d /tmp/rp_$name -o $name --output-format csv -- python3 "$@" > $O/$name.stdout 2> $O/$name.stderr
  find /tmp/rp_$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  echo "done $name"
}
prof bench $R/bench.py
grep '^{' $O/bench.stdout | tail -1 > $O/bench.json
prof jpeg_geoms $R/tests/tools/bench_jpeg_geoms.py
prof hevc_residual $R/tests/tools/bench_hevc_residual.py
prof vp8_residual $R/tests/tools/bench_vp8_residual.py
prof intra_c5 $R/tests/tools/bench_intra_c5.py 6
FRAMES=256 prof vp8_batch256 $R/tests/tools/prof_vp8_batch.py
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_pmc_$c
  rocprofv3 --pmc $c -d /tmp/rp_pmc_$c -o pmc --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > /dev/null 2>&1
  python3 $R/tests/tools/pmc_summary.py /tmp/rp_pmc_$c --kernel k_jpeg420 > $O/pmc_$c.txt
  echo "done pmc $c"
done
pmc() { # name, kernel filter, counters..., then "--", program args
  local name=$1 kern=$2; shift 2
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --pmc "${ctrs[@]}" -d /tmp/rp_$name -o pmc --output-format csv -- python3 "$@" > /dev/null 2>&1
  python3 $R/tests/tools/pmc_summary.py /tmp/rp_$name --kernel $kern >> $O/$name.txt
  echo "done $name"
}
pmc jpeg_geoms_pmc k_jpeg SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- $R/tests/tools/bench_jpeg_geoms.py
export FRAMES=256
pmc vp8_batch256_pmc k_vp8 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- $R/tests/tools/prof_vp8_batch.py
pmc vp8_batch256_pmc k_vp8 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- $R/tests/tools/prof_vp8_batch.py
pmc intra_c5_pmc k_hevc_intra_groups SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- $R/tests/tools/bench_intra_c5.py 6
cd $R
SIZES=1,16,64,256,1024 python3 tests/tools/bench_vp8_batch_sweep.py > $O/vp8_batch_sweep.json 2> /dev/null
NO_CPU=1 python3 tests/tools/bench_hevc_grid.py > $O/hevc_grid.json 2> /dev/null
python3 tests/tools/bench_hevc_tiles.py > $O/hevc_tiles.json 2> /dev/null
FRAMES=256 python3 tests/tools/diag_vp8_batch_waves.py > $O/vp8_batch_waves_256.json 2> /dev/null
FRAMES=16 PRED_WAVES=512,1024 python3 tests/tools/diag_vp8_batch_waves.py > $O/vp8_batch_waves_16.json 2> /dev/null
python3 tests/tools/bench_stages.py --8k > $O/stages_8k.json 2> /dev/null
echo "done stages"
ls -la $O
