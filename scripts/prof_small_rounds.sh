cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_small
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o small -- python3 $GRAFT_REPO_ROOT/scripts/ref_stall_probe.py 10 verify > $OUT/run.log 2>&1
tail -14 $OUT/run.log
python3 - $OUT/small_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
