#!/usr/bin/env python3
"""Known-size reads in the bucket stage's access pattern, to calibrate rocprofv3's FETCH_SIZE for it.

Run ON THE GPU BOX under the same PMC pass as the bench (scripts/profile_round.sh):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/traffic_calibration.py OUT/manifest.json
Each launch of k_gather_probe (csrc/probe.hip) reads exactly 128 B x n_gathers:
    mode 0  one lane per RANDOM 128-byte line (eight 16-byte loads): k_msm_bucket's gather of table entries
    mode 1  one lane per line, consecutive lanes consecutive lines
    mode 2  the same bytes as a wide coalesced stream (the pattern MI355X_MICROARCH.md calibrates: FETCH_SIZE = 1/2)
from a 128-MiB table (the bench's prepared generators: Infinity-Cache sized) and a 1-GiB table (an 8-row CRS
table: far past it).  The manifest lists the launches in order; scripts/summarize_profiles.py matches them with
the counter rows and derives bytes-per-FETCH_SIZE-unit for each pattern.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import verifiable_mpc_amd as vm
    ctx = vm.get_context()
    launches = []
    for table_mib in (128, 1024):
        lines = (table_mib << 20) >> 7
        table = ctx.alloc(lines * 128)
        for mode in (0, 1, 2):
            for rep in range(3):                     # the first launch of a configuration warms the caches / TLB
                n = (1 << 24) + 4096 * len(launches)   # distinct grid sizes: the rows can be matched by size too
                ms = ctx.gather_probe(table.ptr, lines, n, mode, seed=17 + rep)
                launches.append({"kernel": "k_gather_probe", "table_MiB": table_mib, "mode": mode, "rep": rep,
                                 "n_gathers": n, "known_bytes": 128 * n, "ms": ms,
                                 "GBps": 128 * n / (ms * 1e-3) / 1e9 if ms else None})
        del table
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "traffic_manifest.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump({"launches": launches}, f, indent=1)
    for l in launches:
        print(l)


if __name__ == "__main__":
    main()
