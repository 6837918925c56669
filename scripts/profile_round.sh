#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line, the rocprofv3 kernel-trace of the
# same command, and the two PMC passes the HBM-traffic figure needs (separate passes, --pmc only).
#   gpurun --timeout 1500 -- 'bash scripts/profile_round.sh r01d'
# then, back in the build container:  python scripts/summarize_profiles.py r01d
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
tail -c 400 "$OUT/bench_$TAG.json"
cd /tmp
rm -rf "$OUT/prof_$TAG" "$OUT/pmc_fetch_$TAG" "$OUT/pmc_write_$TAG"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -- \
    python3 "$REPO/bench.py" --no-cpu-baseline --no-prove > "$OUT/prof_$TAG.bench.json" 2> "$OUT/prof_$TAG.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$TAG" -- \
    python3 "$REPO/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-prove > /dev/null 2> "$OUT/pmc_fetch_$TAG.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$TAG" -- \
    python3 "$REPO/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-prove > /dev/null 2> "$OUT/pmc_write_$TAG.err"
# keep what travels back small: stats + counter csv only
find "$OUT/prof_$TAG" -name '*kernel_trace.csv' -delete
ls -R "$OUT/prof_$TAG" "$OUT/pmc_fetch_$TAG" "$OUT/pmc_write_$TAG" | head -40
