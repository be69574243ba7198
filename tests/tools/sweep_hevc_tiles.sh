for pct in 67 50 80; do for c in 4 2 3; do
echo "== pct $pct chunks $c"; FFHIP_HEVC_TILE_WAVES_PCT=$pct FFHIP_HEVC_TILE_CHUNKS=$c PICTURES=8 NO_CPU=1 python3 tests/tools/bench_hevc_grid.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['rows']: print(r['pictures'], 'tiles-api chain', r['chain_ms'], 'intra', r['intra_recon_ms'], 'one-call', r['one_call_unpipelined']['chain_ms'], r['one_call_unpipelined']['intra_recon_ms'])"
done; done
