#!/bin/bash
# Round-2 evidence run on the MI355X box: rocprofv3 kernel statistics and PMC passes for bench.py and the stage benches.
# usage (from the repo root on the box): bash tests/tools/collect_profiles.sh   -> gpurun_out/profiles_r2/
set -u
R=$PWD
O=$R/gpurun_out/profiles_r2
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
prof intra_c5 $R/tests/tools/bench_intra_c5.py 6 5
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_pmc_$c
  rocprofv3 --pmc $c -d /tmp/rp_pmc_$c -o pmc --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra > /dev/null 2>&1
  python3 $R/tests/tools/pmc_summary.py /tmp/rp_pmc_$c --kernel k_jpeg420 > $O/pmc_$c.txt
  echo "done pmc $c"
done
rm -rf /tmp/rp_pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/rp_pmc_sq -o pmc --output-format csv -- python3 $R/tests/tools/bench_jpeg_geoms.py > /dev/null 2>&1
python3 $R/tests/tools/pmc_summary.py /tmp/rp_pmc_sq --kernel k_jpeg > $O/jpeg_geoms_pmc.txt
echo "done pmc sq"
cd $R
python3 tests/tools/bench_stages.py --8k > $O/stages_8k.json 2> /dev/null
echo "done stages"
ls -la $O
