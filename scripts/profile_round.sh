#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line, rocprofv3 kernel-trace summaries of the same
# command, of the commitment alone (--no-pipeline: the duration the roofline figure uses) and of the
# Protocol-5 prove in both transcripts, and the two PMC passes the HBM-traffic figure needs (separate passes,
# --pmc only, as MI355X_MICROARCH.md prescribes).
#   gpurun --timeout 2400 -- 'bash scripts/profile_round.sh r02'
# then, back in the build container:  python scripts/summarize_profiles.py r02
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
tail -c 300 "$OUT/bench_$TAG.json"
cd /tmp
rm -rf "$OUT"/prof_${TAG}* "$OUT"/pmc_fetch_$TAG "$OUT"/pmc_write_$TAG "$OUT"/pmc_calib_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -- \
    python3 "$REPO/bench.py" --no-cpu-baseline --no-prove > "$OUT/prof_$TAG.bench.json" 2> "$OUT/prof_$TAG.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_alone" -- \
    python3 "$REPO/bench.py" --no-cpu-baseline --no-prove --no-pipeline > "$OUT/prof_${TAG}_alone.bench.json" 2> "$OUT/prof_${TAG}_alone.err"
for MODE in compact reference; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_prove_$MODE" -- \
      python3 "$REPO/scripts/prove_run.py" $MODE 20 3 > "$OUT/prof_${TAG}_prove_$MODE.json" 2> "$OUT/prof_${TAG}_prove_$MODE.err"
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$TAG" -- \
    python3 "$REPO/bench.py" --steps 4 --warmup 2 --batch 1 --no-cpu-baseline --no-prove > /dev/null 2> "$OUT/pmc_fetch_$TAG.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$TAG" -- \
    python3 "$REPO/bench.py" --steps 4 --warmup 2 --batch 1 --no-cpu-baseline --no-prove > /dev/null 2> "$OUT/pmc_write_$TAG.err"
# the same counter on reads of KNOWN size in the bucket stage's access pattern (csrc/probe.hip): the factor that turns
# FETCH_SIZE into bytes for a one-lane-per-128-byte-line gather (MI355X_MICROARCH.md: calibrate other patterns yourself)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_calib_$TAG" -- \
    python3 "$REPO/scripts/traffic_calibration.py" "$OUT/pmc_calib_$TAG/manifest.json" > "$OUT/pmc_calib_$TAG.log" 2> "$OUT/pmc_calib_$TAG.err"
(cd "$REPO" && git rev-parse --short HEAD 2>/dev/null || cat "$REPO/.revision" 2>/dev/null) > "$OUT/revision_$TAG.txt"
# the sources of the kernel the traffic figure is about, as they were for THIS pass (bench.py compares them with the
# tree it runs from and says so when they differ: the GPU box has no .git)
(cd "$REPO" && sha256sum verifiable_mpc_amd/csrc/msm.hip verifiable_mpc_amd/csrc/msm_sort.hip verifiable_mpc_amd/csrc/ge25519.h verifiable_mpc_amd/csrc/fe25519.h) > "$OUT/sources_$TAG.txt"
# (--batch 1 in the PMC passes: every k_msm_bucket launch is then ONE commitment, the unit the roofline figure uses)
# keep what travels back small: stats + counter csv only
# the timed region's dispatch timeline (which queues run what, when) and the prover's stage table
(cd "$REPO" && bash scripts/pipeline_trace.sh timeline_$TAG 20 36 3 3 > /dev/null 2>&1)
python3 "$REPO/scripts/prove_stages.py" 20 rounds > "$OUT/prove_stages_$TAG.txt" 2>&1
python3 "$REPO/scripts/ref_prove_phases.py" 20 > "$OUT/ref_prove_phases_$TAG.txt" 2>&1
find "$OUT" -name '*kernel_trace.csv' -delete
ls "$OUT" | grep "$TAG" | head -40
