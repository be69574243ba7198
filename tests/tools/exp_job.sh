timeout -k 10 900 python -m pytest tests/test_hevc_intra_gpu.py tests/test_vp8_pred_gpu.py -x -q -m gpu 2>&1 | tail -3
for f in 16 256; do FRAMES=$f PRED_WAVES=1024,2048 timeout -k 10 300 python tests/tools/diag_vp8_batch_waves.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['frames'], d['pred'], json.dumps(d['trace']))"; done
FORMS=0,1 timeout -k 10 600 python tests/tools/bench_intra_c5.py 6 2>/dev/null | tail -1
