for rc in 0 8192 32768 65536; do
echo "REDUCE_CHUNKS=$rc"
VMPC_EXPERIMENTAL=1 VMPC_REDUCE_CHUNKS=$rc ROWS=13 python scripts/rows20_probe.py 2>&1 | grep -v amdgpu
VMPC_EXPERIMENTAL=1 VMPC_REDUCE_CHUNKS=$rc python bench.py --steps 24 --warmup 3 --no-prove --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  bench:', round(l['ms_per_step'],4), l['config']['timing']['ms_per_step_of_each_repeat'], l['checked'])"
done
