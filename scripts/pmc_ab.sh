#!/bin/bash
# FETCH_SIZE of k_msm_bucket under two library builds (GPU box): bash scripts/pmc_ab.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab_$(basename $L .so)
  rm -rf $OUT; mkdir -p $OUT
  VMPC_LIB_PATH=$GRAFT_REPO_ROOT/$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --batch 1 --no-cpu-baseline --no-prove > /dev/null 2> $OUT.err
  python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True))[-1]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_msm_bucket(") and r["Counter_Name"] == "FETCH_SIZE"]
print("$L", "k_msm_bucket launches", len(v), "FETCH_SIZE KiB avg %.0f" % (sum(v) / len(v)))
PY
done
