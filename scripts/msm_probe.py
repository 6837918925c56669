#!/usr/bin/env python3
"""Developer probe: MSM stage timings on the GPU box (not part of the product or tests).

usage: python scripts/msm_probe.py [log2n ...]
"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from verifiable_mpc_amd import _native as nat

ELL = 2**252 + 27742317777372353535851937790883648493
BASE = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
        46316835694926478169428394003475163141307993866256225615783033603165251855960)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
    print(nat.backend_info())
    ctx = nat.Context(0)
    base = np.frombuffer(BASE[0].to_bytes(32, "little") + BASE[1].to_bytes(32, "little"), dtype=np.uint8)
    dbase = ctx.upload(base)
    rng = np.random.default_rng(1)
    for lg in sizes:
        n = 1 << lg
        raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        raw[:, 31] &= 0x0f                       # < 2^252 < l
        exps = raw.copy()
        sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        sc[:, 31] &= 0x0f
        dexp, dsc = ctx.upload(exps), ctx.upload(sc)
        dpts = ctx.alloc(64 * n)
        t0 = time.time()
        ctx.repeat(dbase.ptr, 1, True, dexp.ptr, n, False, None, dpts.ptr)
        ctx.sync()
        t_gen = time.time() - t0
        out = ctx.alloc(64)
        ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, out.ptr)   # warm-up (+ workspace growth)
        ctx.sync()
        ctx.profile(True)
        ctx.profile_read(reset=True)
        reps = 5
        t0 = time.time()
        for _ in range(reps):
            ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, out.ptr)
        ctx.sync()
        dt = (time.time() - t0) / reps
        prof = ctx.profile_read(reset=True)
        ctx.profile(False)
        t0 = time.time()
        for _ in range(reps):
            ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, out.ptr)
        ctx.sync()
        dt_np = (time.time() - t0) / reps
        # property check: sum s_i e_i * B
        e_int = nat.array_to_ints(exps)
        s_int = nat.array_to_ints(sc)
        tot = sum(a * b for a, b in zip(e_int, s_int)) % ELL
        dtot = ctx.upload(nat.ints_to_array([tot]))
        chk = ctx.alloc(64)
        ctx.repeat(dbase.ptr, 1, True, dtot.ptr, 1, False, None, chk.ptr)
        ctx.sync()
        ok = bool((ctx.download(chk.ptr, 64) == ctx.download(out.ptr, 64)).all())
        print(f"n=2^{lg}: gen {t_gen*1e3:.1f} ms; msm {dt*1e3:.3f} ms profiled / {dt_np*1e3:.3f} ms plain "
              f"-> {n/dt_np/1e6:.1f} M sm/s; correct={ok}")
        for k, (ms, cnt) in prof.items():
            print(f"    {k:18s} {ms/max(cnt,1)*1e3:10.1f} us x{cnt//reps}")
        # fold timing
        half = n // 2
        op = ctx.alloc(64 * half)
        ctx.fold(dpts.ptr, dpts.ptr + 64 * half, True, tot, half, None, op.ptr)
        ctx.sync()
        t0 = time.time()
        ctx.fold(dpts.ptr, dpts.ptr + 64 * half, True, tot, half, None, op.ptr)
        ctx.sync()
        print(f"    fold half=2^{lg-1}: {(time.time()-t0)*1e3:.2f} ms")


if __name__ == "__main__":
    main()
