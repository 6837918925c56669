import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import verifiable_mpc_amd as vm
ctx = vm.get_context()
rng = np.random.default_rng(5)
group = vm.EllipticCurve("Ed25519", "projective")
half = 1 << 15
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, 2 * half)), keep_proj=True)
c = int.from_bytes(bench.rand_scalars(rng, 1)[0].tobytes(), "little")
vm.PointVector.TEXT_FIRST_SLICE, vm.PointVector.TEXT_SLICED_FROM = 1 << 14, 1 << 14
for rep in range(3):
    ctx.sync(); time.sleep(0.01)
    out = g[:half].fold(g[half:], c, stream_text=True)
    for p in out.text_chunks():
        pass
