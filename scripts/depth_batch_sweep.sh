for d in 1 2 3; do for b in 1 2 3 4 6 8; do
  python bench.py --depth $d --batch $b --steps 24 --warmup 3 --no-cpu-baseline --no-prove 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth',d['config']['launches_in_flight'],'batch',d['config']['commitments_per_launch'],'ms/step %.4f'%d['ms_per_step'],'varbase %.4f'%d['variable_base']['ms_per_step'],'alone %.3f'%d['alone']['ms_per_commitment'])
"
done; done
