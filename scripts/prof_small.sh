#!/bin/bash
# rocprofv3 kernel stats of the compact prover at N = 2^$1 (scratch: gpurun_out/$2)
k=${1:-12}; out=${2:-gpurun_out/small}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o p -- python3 $R/scripts/prove_run.py compact $k ${REPS:-20} 2>/dev/null | grep "^{" > $R/$out/run.log
f=$(find $R/$out/prof -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $R/$out/stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{r['Name'][:40]:40s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} tot_ms {float(r['TotalDurationNs'])/1e6:9.2f} min_us {float(r['MinNs'])/1e3:8.1f}")
PY
rm -rf $R/$out/prof
