#!/bin/bash
# quick A/B of tuning knobs on the GPU box: bash scripts/bench_quick.sh TAG "ENV=..." ["ENV=..." ...]
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --no-cpu-baseline --no-prove > gpurun_out/$TAG/b$i.json 2> gpurun_out/$TAG/b$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/b$i.json"))
a=d["alone"]["stages_us"]
print("$envs", "| step %.4f ms (other form %.4f) alone %.3f | bucket %.0f sort %.0f part %.0f reduce %.0f" % (d["ms_per_step"], list(d.values())[[k for k in d].index("variable_base" if "variable_base" in d else "prepared_generators")]["ms_per_step"], d["alone"]["ms_per_commitment"], a["msm_bucket"], a["msm_sort"], a["msm_part"], a["msm_reduce"]))
PY
done
