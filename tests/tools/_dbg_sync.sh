set -u
for t in no_dri dri_per_mcu_row no_dri,dri_per_mcu_row; do
F1_TAGS=$t python bench.py --extras f1 --no-cpu 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
f=d['configs']['f1']
print({k:(v['ms'],v['entropy_gpu']) for k,v in f.items()})"
done
