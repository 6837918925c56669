#!/bin/bash
# Runs ON THE GPU BOX: the probes DESIGN.md cites, outputs under gpurun_out/probes_<tag>/ (copy to profiles/ afterwards)
TAG=${1:-rXX}
OUT=gpurun_out/probes_$TAG
mkdir -p $OUT
python scripts/rows_probe.py 20 2>&1 | grep "rows=" > $OUT/rows_probe.txt
python scripts/pair_probe.py 8 2>&1 | grep "rows=" > $OUT/pair_probe.txt
python scripts/batch_probe.py 20 2>&1 | grep "K=" > $OUT/batch_probe.txt
python scripts/prove_stages.py 20 rounds 2>&1 | grep -v amdgpu > $OUT/prove_stages.txt
python scripts/loop_probe2.py 2>&1 | grep -v amdgpu > $OUT/loop_probe2.txt
bash scripts/depth_batch_sweep.sh > $OUT/depth_batch_sweep.txt 2>&1
bash scripts/rows_throughput.sh > $OUT/rows_throughput.txt 2>&1
for k in 10 12 14 16 18 20; do echo -n "N=2^$k: "; python scripts/prove_run.py compact $k 5 | tail -1; done > $OUT/prove_sizes.txt 2>&1
python scripts/pinocchio_probe.py 18 2>&1 | grep -v amdgpu > $OUT/pinocchio_probe.txt
ls -la $OUT
