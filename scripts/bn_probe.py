#!/usr/bin/env python3
"""Developer probe: BN-256 G1/G2 MSM timing (BASELINE config 5 sizes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import _native
ctx = vm.get_context()
G1 = (1).to_bytes(32, "little") + (65000549695646603732796438742359905742825358107623003571877145026864184071783 - 2).to_bytes(32, "little")
G2v = (64746500191241794695844075326670126197795977525365406531717464316923369116492,
       21167961636542580255011770066570541300993051739349375019639421053990175267184,
       17778617556404439934652658462602675281523610326338642107814333856843981424549,
       20666913350058776956210519119118544732556678129809273996262322366050359951122)
G2 = b"".join(v.to_bytes(32, "little") for v in G2v)
rng = np.random.default_rng(1)
for group, gen, width in ((1, G1, 64), (2, G2, 128)):
    for lg in [int(a) for a in sys.argv[1:]] or [12, 16, 18]:
        n = 1 << lg
        # points: small multiples of the generator made by n one-term MSMs would be slow; use the
        # generator repeated with random scalars (timing only; correctness is in tests/)
        pts = np.tile(np.frombuffer(gen, np.uint8), (n, 1))
        sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x7f
        dp, ds, out = ctx.upload(pts), ctx.upload(sc), ctx.alloc(width)
        ctx.bn256_msm(group, ds.ptr, dp.ptr, n, out.ptr); ctx.sync()
        ctx.profile(True); ctx.profile_read(True)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.bn256_msm(group, ds.ptr, dp.ptr, n, out.ptr)
        ctx.sync(); dt = (time.perf_counter() - t0) / 3
        prof = ctx.profile_read(True); ctx.profile(False)
        print(f"G{group} n=2^{lg}: {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} M sm/s  " +
              " ".join(f"{k}={ms/max(c,1)*1e3:.0f}us" for k, (ms, c) in prof.items() if k.startswith("bn_")))
        want = ctx.download(out.ptr, width).tobytes()
        t0 = time.perf_counter()
        table = ctx.bn256_table_build(group, dp.ptr, n); ctx.sync()
        t_build = time.perf_counter() - t0
        out2 = ctx.alloc(width)
        ctx.bn256_table_msm(group, table.ptr, n, ds.ptr, n, out2.ptr, None); ctx.sync()
        same = ctx.download(out2.ptr, width).tobytes() == want
        ctx.profile(True); ctx.profile_read(True)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.bn256_table_msm(group, table.ptr, n, ds.ptr, n, out2.ptr, None)
        ctx.sync(); dt = (time.perf_counter() - t0) / 3
        prof = ctx.profile_read(True); ctx.profile(False)
        print(f"   table: {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} M sm/s same={same} build {t_build*1e3:.0f} ms {table.nbytes>>20} MiB  " +
              " ".join(f"{k}={ms/max(c,1)*1e3:.0f}us" for k, (ms, c) in prof.items() if k.startswith("bn_")))
        del table
