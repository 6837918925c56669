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
                # "void k_msm_bucket<256>(unsigned int const*, ...)" -> "k_msm_bucket": templated kernels carry their
                # return type and arguments in the profiler's name
                name = r["Kernel_Name"].split("(")[0]
                name = name[5:] if name.startswith("void ") else name
                name = name.split("<")[0]
                rows.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in rows.items()}


def calibration(out, tag):
    """bytes per FETCH_SIZE KiB for the probe's three patterns and two table sizes (scripts/traffic_calibration.py)"""
    d = os.path.join(out, f"pmc_calib_{tag}")
    man = os.path.join(d, "manifest.json")
    paths = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not os.path.exists(man) or not paths:
        return None
    with open(man) as f:
        launches = json.load(f)["launches"]
    rows = []
    with open(paths[-1]) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == "FETCH_SIZE" and r["Kernel_Name"].startswith("k_gather_probe"):
                rows.append((int(r.get("Dispatch_Id", len(rows))), float(r["Counter_Value"]), r.get("Grid_Size")))
    rows.sort()
    if len(rows) != len(launches):
        return {"error": f"{len(rows)} counter rows for {len(launches)} launches"}
    names = {0: "gather_random_lines", 1: "lines_in_order", 2: "coalesced_stream"}
    res = {}
    for l, (_, kib, grid) in zip(launches, rows):
        if l["rep"] == 0:
            continue                            # warm-up launch of the configuration
        key = f"{names[l['mode']]}_{l['table_MiB']}MiB"
        e = res.setdefault(key, {"known_bytes": l["known_bytes"], "FETCH_SIZE_KiB": [], "ms": [], "GBps": []})
        e["FETCH_SIZE_KiB"].append(kib)
        e["ms"].append(l["ms"])
        e["GBps"].append(l["GBps"])
    for e in res.values():
        kib = sum(e["FETCH_SIZE_KiB"]) / len(e["FETCH_SIZE_KiB"])
        e["FETCH_SIZE_KiB_avg"] = kib
        e["bytes_per_FETCH_SIZE_byte"] = e["known_bytes"] / (kib * 1024.0) if kib else None
        e["GBps_avg"] = sum(e["GBps"]) / len(e["GBps"])
    return res


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
    for src, name in ((os.path.join(out, f"timeline_{tag}", "timeline.txt"), "timeline.txt"),
                      (os.path.join(out, f"prove_stages_{tag}.txt"), "prove_stages.txt"),
                      (os.path.join(out, f"ref_prove_phases_{tag}.txt"), "ref_prove_phases.txt")):
        if os.path.exists(src):
            shutil.copy(src, os.path.join(prof, f"{tag}_{name}"))
    fetch = counter_avgs(os.path.join(out, f"pmc_fetch_{tag}"), "FETCH_SIZE")
    write = counter_avgs(os.path.join(out, f"pmc_write_{tag}"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, f_n = fetch.get(k, (0.0, 0))
        w_kib, w_n = write.get(k, (0.0, 0))
        kernels[k] = {"FETCH_SIZE_KiB_avg": f_kib, "WRITE_SIZE_KiB_avg": w_kib,
                      "hbm_bytes_per_launch_corrected": (2.0 * f_kib + w_kib) * 1024.0,
                      "launches_FETCH_SIZE": f_n, "launches_WRITE_SIZE": w_n}
    calib = calibration(out, tag)
    calibrated = {}
    rev = None
    if os.path.exists(os.path.join(out, f"revision_{tag}.txt")):
        with open(os.path.join(out, f"revision_{tag}.txt")) as f:
            rev = f.read().strip() or None
    if calib and "error" not in calib and "k_msm_bucket" in kernels:
        # k_msm_bucket's reads are one-lane-per-line gathers from the 128-MiB prepared-generator table plus the
        # sorted index stream (4 B per entry, in order): apply the factor measured for the gather in this pass
        key = "gather_random_lines_128MiB"
        fac = calib[key]["bytes_per_FETCH_SIZE_byte"]
        kb = kernels["k_msm_bucket"]
        n, windows = 1 << 20, 16
        calibrated["k_msm_bucket"] = {
            "fetch_factor": fac, "factor_from": key,
            "hbm_bytes_per_launch": (fac * kb["FETCH_SIZE_KiB_avg"] + kb["WRITE_SIZE_KiB_avg"]) * 1024.0,
            "structural_bytes": windows * n * (128 + 4) + windows * (1 << 15) * 160,
            "structural_note": "16 windows x 2^20 terms x (128-B table line + 4-B sorted index) read, "
                               "16 x 2^15 buckets x 160 B written"}
    sources = {}
    if os.path.exists(os.path.join(out, f"sources_{tag}.txt")):
        with open(os.path.join(out, f"sources_{tag}.txt")) as f:
            for ln in f:
                h, name = ln.split()
                sources[name] = h
    summary = {"revision": rev, "kernel_sources_sha256": sources, "calibration": calib, "calibrated": calibrated, "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 4 --warmup 2 "
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
