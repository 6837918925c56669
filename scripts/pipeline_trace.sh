#!/bin/bash
# kernel trace of the bench's timed region + its timeline summary -> gpurun_out/$1/{run.json,timeline.txt}
#   gpurun -- 'bash scripts/pipeline_trace.sh tl_base [log2n steps batch depth]'
tag=${1:-tl}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof
rocprofv3 --kernel-trace --output-format csv -d $out/prof -- python3 $R/scripts/pipeline_run.py "$@" > $out/run.json 2> $out/run.err
f=$(find $out/prof -name '*kernel_trace.csv' | head -1)
python3 $R/scripts/timeline.py "$f" > $out/timeline.txt 2>&1
rm -rf $out/prof
cat $out/run.json; head -40 $out/timeline.txt
