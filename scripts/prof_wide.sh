#!/bin/bash
# per-kernel durations of one wide-window (13-row) commitment at 2^20, alone and as passes of three: rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_wide
rm -rf $OUT; mkdir -p $OUT
ROWS=${ROWS:-13} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o wide -- python3 $GRAFT_REPO_ROOT/scripts/rows20_probe.py > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r06_wide_kernel_stats.csv
cat $OUT/run.log | tail -4
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
