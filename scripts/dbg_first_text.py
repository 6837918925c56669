import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import verifiable_mpc_amd as vm
ctx = vm.get_context()
rng = np.random.default_rng(5)
group = vm.EllipticCurve("Ed25519", "projective")
for log_half in (19, 16, 15):
    half = 1 << log_half
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, 2 * half)), keep_proj=True)
    c = int.from_bytes(bench.rand_scalars(rng, 1)[0].tobytes(), "little")
    for first, sliced_from in ((1 << 16, 1 << 17), (1 << 14, 1 << 14), (1 << 13, 1 << 14), (1 << 12, 1 << 14)):
        vm.PointVector.TEXT_FIRST_SLICE, vm.PointVector.TEXT_SLICED_FROM = first, sliced_from
        res = []
        for rep in range(3):
            ctx.sync(); time.sleep(0.01)
            t0 = time.perf_counter()
            out = g[:half].fold(g[half:], c, stream_text=True)
            t1 = time.perf_counter()
            it = out.text_chunks()
            p = next(it); t2 = time.perf_counter()
            n = len(p)
            for p in it:
                n += len(p)
            t3 = time.perf_counter()
            res.append((t1 - t0, t2 - t0, t3 - t0))
        e, f, a = min(res)
        print(f"half 2^{log_half} first {first:6d} sliced_from {sliced_from:7d}: enqueue {e*1e3:.2f} ms, first text at {f*1e3:.2f} ms, all {a*1e3:.2f} ms ({n/1e6:.0f} MB)", flush=True)
