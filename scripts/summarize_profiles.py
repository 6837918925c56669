#!/usr/bin/env python3
"""Turn gpurun_out/{bench,prof,pmc_fetch,pmc_write}_<tag> into the committed profiles/<tag>_* files.

HBM bytes per launch follow MI355X_MICROARCH.md ("HBM" section): FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes (KiB); on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes,
so it is doubled; WRITE_SIZE is taken as reported.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avgs(dirname, counter):
    rows = {}
    paths = sorted(glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True),
                   key=os.path.getmtime)
    for path in paths[-1:]:          # a tag profiled twice leaves two files: the newest wins
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                name = r["Kernel_Name"].split("(")[0]
                rows.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in rows.items()}


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(out, f"bench_{tag}.json"), os.path.join(prof, f"{tag}_bench.json"))
    for sub, name in ((f"prof_{tag}", "msm_n2^20_kernel_stats.csv"),
                      (f"prof_{tag}_alone", "msm_n2^20_alone_kernel_stats.csv"),
                      (f"prof_{tag}_prove_compact", "prove_compact_kernel_stats.csv"),
                      (f"prof_{tag}_prove_reference", "prove_reference_kernel_stats.csv")):
        stats = sorted(glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True),
                       key=os.path.getmtime)
        if stats:
            shutil.copy(stats[-1], os.path.join(prof, f"{tag}_{name}"))
    for sub, name in ((f"prof_{tag}.bench.json", "bench_under_rocprof.json"),
                      (f"prof_{tag}_alone.bench.json", "bench_alone_under_rocprof.json"),
                      (f"prof_{tag}_prove_compact.json", "prove_compact_under_rocprof.json"),
                      (f"prof_{tag}_prove_reference.json", "prove_reference_under_rocprof.json")):
        if os.path.exists(os.path.join(out, sub)):
            shutil.copy(os.path.join(out, sub), os.path.join(prof, f"{tag}_{name}"))
    fetch = counter_avgs(os.path.join(out, f"pmc_fetch_{tag}"), "FETCH_SIZE")
    write = counter_avgs(os.path.join(out, f"pmc_write_{tag}"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, f_n = fetch.get(k, (0.0, 0))
        w_kib, w_n = write.get(k, (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KiB_avg": f_kib, "WRITE_SIZE_KiB_avg": w_kib,
                      "hbm_bytes_per_launch_corrected": (2.0 * f_kib + w_kib) * 1024.0,
                      "launches_FETCH_SIZE": f_n, "launches_WRITE_SIZE": w_n}
    summary = {"command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 4 --warmup 2 "
                          "--batch 1 --no-cpu-baseline --no-prove (one pass per counter, one commitment per launch; "
                          "scripts/profile_round.sh)",
               "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE tallies 128-B "
                             "requests at 64 B, MI355X_MICROARCH.md HBM section)",
               "kernels": kernels}
    with open(os.path.join(prof, f"{tag}_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote", [p for p in os.listdir(prof) if p.startswith(tag)])
    if "k_msm_bucket" in kernels:
        print("k_msm_bucket HBM bytes/launch:", kernels["k_msm_bucket"]["hbm_bytes_per_launch_corrected"])


if __name__ == "__main__":
    main()
