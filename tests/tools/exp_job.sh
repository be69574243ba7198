timeout -k 10 900 python -m pytest tests/test_hevc_intra_gpu.py tests/test_handoff_stress_gpu.py -x -q -m gpu 2>&1 | tail -3
FORMS=0,1 timeout -k 10 600 python tests/tools/bench_intra_c5.py 6 2>/dev/null | tail -1
