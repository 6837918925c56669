#!/usr/bin/env python3
"""Developer probe: fixed-base table MSM vs variable-base MSM (correctness + stage timings)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from verifiable_mpc_amd import _native as nat

BASE = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
        46316835694926478169428394003475163141307993866256225615783033603165251855960)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [10, 16, 20]
    ctx = nat.Context(0)
    base = np.frombuffer(BASE[0].to_bytes(32, "little") + BASE[1].to_bytes(32, "little"), dtype=np.uint8)
    dbase = ctx.upload(base)
    rng = np.random.default_rng(1)
    for lg in sizes:
        n = (1 << lg) - 1
        n_extra = 2
        exps = rng.integers(0, 256, size=(n + n_extra, 32), dtype=np.uint8)
        exps[:, 31] &= 0x0f
        sc = rng.integers(0, 256, size=(n + n_extra, 32), dtype=np.uint8)
        sc[:, 31] &= 0x0f
        dexp, dsc = ctx.upload(exps), ctx.upload(sc)
        dpts = ctx.alloc(64 * (n + n_extra))
        ctx.repeat(dbase.ptr, 1, True, dexp.ptr, n + n_extra, False, None, dpts.ptr)
        ctx.sync()
        for rows in [int(r) for r in os.environ.get("ROWS", "16").split(",")]:
            t0 = time.time()
            table = ctx.msm_table_build(dpts.ptr, n, dpts.ptr + 64 * n, n_extra, rows)
            ctx.sync()
            t_build = time.time() - t0
            out_v, out_t = ctx.alloc(64), ctx.alloc(64)
            print(f"rows={rows}")
            for m in (n, n // 2 + 1, 1, 0):
                ctx.msm(dsc.ptr, dpts.ptr, m, dsc.ptr + 32 * n, dpts.ptr + 64 * n, n_extra, None, out_v.ptr)
                ctx.msm_table(table.ptr, n, n_extra, dsc.ptr, m, dsc.ptr + 32 * n, None, out_t.ptr, rows)
                ctx.sync()
                same = ctx.download(out_v.ptr, 64).tobytes() == ctx.download(out_t.ptr, 64).tobytes()
                print(f"n=2^{lg}-1 m={m}: table == variable-base: {same}")
            ctx.msm_table(table.ptr, n, n_extra, dsc.ptr, n, None, None, out_t.ptr, rows)
            ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, out_v.ptr)
            ctx.sync()
            print("  no extras:", ctx.download(out_v.ptr, 64).tobytes() == ctx.download(out_t.ptr, 64).tobytes())
            reps = 5
            for name, fn in (("variable", lambda: ctx.msm(dsc.ptr, dpts.ptr, n, dsc.ptr + 32 * n, dpts.ptr + 64 * n, n_extra, None, out_v.ptr)),
                             ("table", lambda: ctx.msm_table(table.ptr, n, n_extra, dsc.ptr, n, dsc.ptr + 32 * n, None, out_t.ptr, rows))):
                fn(); ctx.sync()
                t0 = time.time()
                for _ in range(reps):
                    fn()
                ctx.sync()
                dt = (time.time() - t0) / reps
                ctx.profile(True); ctx.profile_read(reset=True)
                fn(); ctx.sync()
                prof = ctx.profile_read(reset=True); ctx.profile(False)
                st = " ".join(f"{k[4:]}={ms*1e3:.0f}" for k, (ms, cnt) in prof.items())
                print(f"  {name:9s} {dt*1e3:7.3f} ms  {st}")
            print(f"  table build {t_build*1e3:.1f} ms, {table.nbytes/2**20:.0f} MiB")


if __name__ == "__main__":
    main()
