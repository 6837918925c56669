#!/bin/bash
# A/B on ONE box: compact prover at N = 2^$1 with and without the environment setting "$2" (alternating runs)
k=${1:-20}; setting=$2; reps=${3:-3}
for i in $(seq $reps); do
  echo -n "base   : "; python3 scripts/prove_run.py compact $k 5 2>/dev/null | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sorted(d["prove_ms"])[len(d["prove_ms"])//2])'
  echo -n "$setting: "; env $setting python3 scripts/prove_run.py compact $k 5 2>/dev/null | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sorted(d["prove_ms"])[len(d["prove_ms"])//2])'
done
