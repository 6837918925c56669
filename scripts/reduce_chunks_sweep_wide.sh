# reduction shapes for the wide-window pass of three (scripts/rows20_probe.py stages + bench.py headline)
for cfg in "0 1" "32768 0" "32768 1" "16384 0" "8192 0"; do set -- $cfg
echo "REDUCE_CHUNKS=$1 REDUCE_TREE=$2"
VMPC_EXPERIMENTAL=1 VMPC_REDUCE_CHUNKS=$1 VMPC_REDUCE_TREE=$2 ROWS=13 python scripts/rows20_probe.py 2>&1 | grep -E "stages of a pass|alone"
VMPC_EXPERIMENTAL=1 VMPC_REDUCE_CHUNKS=$1 VMPC_REDUCE_TREE=$2 python bench.py --steps 24 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  bench:', round(l['ms_per_step'],4), l['config']['timing']['ms_per_step_of_each_repeat'], l['checked'])"
done
