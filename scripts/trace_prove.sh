#!/bin/bash
# per-launch kernel durations of ONE compact prove at N = 2^$1, in launch order (scratch: gpurun_out/$2/trace.txt)
k=${1:-20}; out=${2:-gpurun_out/trace}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --kernel-trace --output-format csv -d $R/$out/prof -- python3 $R/scripts/prove_run.py compact $k 2 2>/dev/null | grep '^{' > $R/$out/run.log
f=$(find $R/$out/prof -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $R/$out/trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last prove = everything after the last-but-one k_fold_jump ... simpler: print the tail that starts at the last k_p4_fold_dots with fold=0 -> take last 700 launches
tail = rows[-700:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = t0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:44]}")
    prev_end = e
PY
rm -rf $R/$out/prof
