#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one bench pass (separate --pmc run, no tracing besides kernel names).
#   gpurun -- 'bash scripts/pmc_kernel.sh TAG "SQ_WAVES SQ_WAVE_CYCLES ..."'
TAG=$1; CTRS=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prove --no-pipeline > /dev/null 2> $OUT.err
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if k.startswith(("k_sort", "k_msm", "void k_sort", "void k_scan")):
        print("%-30s" % k, "  ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
