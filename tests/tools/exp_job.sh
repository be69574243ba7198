for rep in 1 2; do for v in A B; do
  echo "== $v rep $rep"
  FFHIP_LIB=libffpic_hip_$v.so timeout -k 10 300 python tests/tools/bench_intra_c5.py 6 2>/dev/null | tail -1
  FFHIP_LIB=libffpic_hip_$v.so NO_CPU=1 timeout -k 10 300 python tests/tools/bench_hevc_grid.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print([(r['pictures'], r['intra_recon_ms']) for r in d['rows']])"
done; done
for w in 640 1024 1280 1600 2048; do echo "== B waves $w"; FFHIP_HEVC_INTRA_WAVES=$w FFHIP_LIB=libffpic_hip_B.so NO_CPU=1 timeout -k 10 300 python tests/tools/bench_hevc_grid.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print([(r['pictures'], r['intra_recon_ms']) for r in d['rows']])"; done
