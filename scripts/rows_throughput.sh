#!/bin/bash
# bench.py throughput against the rows of the resident generator table and the commitments per pass (GPU box)
for r in 1 2 4 8; do for b in 3 6; do
  python bench.py --table-rows $r --batch $b --no-cpu-baseline --no-prove 2>/dev/null | R=$r B=$b python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows', os.environ['R'], 'batch', os.environ['B'], 'ms/step %.4f' % d['ms_per_step'], 'G/s %.3f' % (d['value'] / 1e9),
      'alone %.3f' % d['alone']['ms_per_commitment'], d['checked'])"
done; done
