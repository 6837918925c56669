for cfg in "3 0" "3 2" "4 2" "2 3" "6 1" "4 3"; do set -- $cfg
python bench.py --steps 24 --warmup 3 --table-rows 13 --no-prove --no-cpu-baseline --batch $1 --depth $2 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $1 depth $2:', round(l['ms_per_step'],4), l['config']['timing']['ms_per_step_of_each_repeat'], l['checked'])"
done
