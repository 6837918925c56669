#!/bin/bash
# kernel-by-kernel timeline of the small rounds of a reference-transcript prove / verify (N = 2^k): rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
K=${1:-12}; WHAT=${2:-prove}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_small
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o small -- python3 $GRAFT_REPO_ROOT/scripts/ref_stall_probe.py $K $WHAT > $OUT/run.log 2>&1
grep -E "^prove|^verify|sums" $OUT/run.log
python3 - $OUT/small_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-150:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
for r in last:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(st - t0) / 1e3:9.1f} us  +{(st - prev_end) / 1e3:7.1f} gap  {(en - st) / 1e3:8.1f} us  q{r.get("Queue_Id", "?"):>3}  {r["Kernel_Name"][:60]}')
    prev_end = max(prev_end, en)
PY
