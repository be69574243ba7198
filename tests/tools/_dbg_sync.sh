set -u
python3 -c "
import ctypes
from ffpic_amd import capi
L=capi.require_device()
L.ffhip_debug_numa_node.restype=ctypes.c_int
print('gpu numa node', L.ffhip_debug_numa_node())
import os
print('cpus allowed', sorted(os.sched_getaffinity(0))[:40])
"
for v in 1 0 1 0; do
FFHIP_NUMA=$v python bench.py --extras f1 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
f=d['configs']['f1']
print('numa $v', {k:(v['ms'],v['device_pipeline_ms']) for k,v in f.items()})"
done
