for t in 1 0 1 0; do echo "REDUCE_TREE=$t"
VMPC_EXPERIMENTAL=1 VMPC_REDUCE_TREE=$t ROWS=13 python scripts/rows20_probe.py 2>&1 | grep -E "stages of a pass"
VMPC_EXPERIMENTAL=1 VMPC_REDUCE_TREE=$t python bench.py --steps 24 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  bench:', round(l['ms_per_step'],4), l['config']['timing']['ms_per_step_of_each_repeat'], l['checked'])"
done
