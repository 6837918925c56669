"""Where the Protocol-5 prelude (everything before the halving rounds) spends its time at N = 2^k, compact transcript."""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import compressed_pivot as cp, pivot
def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = vm.get_context()
rng = np.random.default_rng(3)
n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective")
gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)), keep_proj=False)
h, kk = group.generator, vm.Ed25519Point.repeat(group.generator, 12345)
gens = {"g": g, "h": h, "k": kk}
g.precompute([h, kk])
x = vm.ScalarVector.from_array(rs(rng, n))
L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x))
P = vm.pivot.vector_commitment(x, 777, g, h)
cp.generators_digest(gens)
order = gf.order
for rep in range(3):
    r = vm.ScalarVector.from_array(rs(rng, n)); rho = 5
    ctx.sync()
    T = [("start", time.perf_counter())]
    mark = lambda name: T.append((name, time.perf_counter()))
    Lr, yr = pivot.affine_to_linear(L, y, n); mark("affine_to_linear")
    cp._form_digest_begin(Lr); mark("form_digest_begin")
    pending = pivot._commit_launch(r, rho, g, h, g.ctx); mark("A launch")
    Lr._form_digest = cp._form_digest(Lr); mark("form digest (host part)")
    t = Lr(r); mark("t = L(r)")
    A = pending.result(); mark("A result")
    t = gf(t)
    c0, c1, seed = cp._p5_challenges("compact", order, gens, t, A, P, Lr, yr); mark("challenges")
    z = x.axpy(c0, r); mark("z axpy")
    z_hat = z + [gf(c0 * 777 + rho)]; mark("z_hat concat")
    g_hat = g + [h]; mark("g_hat concat")
    L_tilde = cp._extend_form(Lr, c1); mark("L_tilde")
    tr = cp._p5_setup(gens, kk, seed, "compact", order); mark("p5_setup")
    ctx.sync(); mark("sync")
    t0 = time.perf_counter()
    proof = cp.protocol_4_prover(g_hat, kk, cp._LazyQ(A, P, kk, c0, int(c1 * (c0 * yr + t)), order), L_tilde, z_hat, gf, {}, transcript=tr)
    ctx.sync()
    rounds_ms = (time.perf_counter() - t0) * 1e3
    if rep == 2:
        for (a, ta), (b, tb) in zip(T, T[1:]):
            print(f"  {b:28s} {(tb - ta) * 1e3:7.3f} ms")
        print(f"  prelude total {(T[-1][1] - T[0][1]) * 1e3:.3f} ms, rounds {rounds_ms:.3f} ms")
