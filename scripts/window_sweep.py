#!/usr/bin/env python3
"""Developer probe: MSM time vs window width c (vmpc_ctx_set_window) on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from verifiable_mpc_amd import _native as nat

BASE = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
        46316835694926478169428394003475163141307993866256225615783033603165251855960)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 20]
    ctx = nat.Context(0)
    base = np.frombuffer(BASE[0].to_bytes(32, "little") + BASE[1].to_bytes(32, "little"), dtype=np.uint8)
    dbase = ctx.upload(base)
    rng = np.random.default_rng(1)
    for lg in sizes:
        n = 1 << lg
        exps = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        exps[:, 31] &= 0x0f
        sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        sc[:, 31] &= 0x0f
        dexp, dsc = ctx.upload(exps), ctx.upload(sc)
        dpts = ctx.alloc(64 * n)
        ctx.repeat(dbase.ptr, 1, True, dexp.ptr, n, False, None, dpts.ptr)
        out = ctx.alloc(64)
        ref = None
        for c in [0] + list(range(int(os.environ.get("CMIN", "8")), 17)):
            ctx.set_window(c)
            ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, out.ptr)
            ctx.sync()
            res = ctx.download(out.ptr, 64).tobytes()
            ref = ref or res
            reps = 5
            t0 = time.time()
            for _ in range(reps):
                ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, out.ptr)
            ctx.sync()
            dt = (time.time() - t0) / reps
            ctx.profile(True)
            ctx.profile_read(reset=True)
            ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, out.ptr)
            ctx.sync()
            prof = ctx.profile_read(reset=True)
            ctx.profile(False)
            st = " ".join(f"{k[4:]}={ms*1e3:.0f}" for k, (ms, cnt) in prof.items())
            print(f"n=2^{lg} c={c:2d}: {dt*1e3:7.3f} ms  same={res == ref}  {st}")
        ctx.set_window(0)


if __name__ == "__main__":
    main()
