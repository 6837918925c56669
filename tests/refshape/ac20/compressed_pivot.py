"""Stand-in for verifiable_mpc/ac20/compressed_pivot.py (AC20 Protocols 4 and 5): the reference's names, signatures and
proof-dict keys over own generic-group CPU code (a loop where the reference recurses).  After
`verifiable_mpc_amd.install("tests.refshape.ac20")` these four names dispatch on the group; what is written here is
what a QuadraticResidues call falls through to.  Everything goes through `pivot.<name>` - the module attribute - as the
reference does (compressed_pivot.py:41-42,59,110,125-130), so that rebinding pivot's names reaches this code too.
"""
from random import SystemRandom

from mpyc.fingroups import EllipticCurvePoint

from . import pivot

prng = SystemRandom()
TAG = "First hash of compressed pivot"


def _norm(*elements):
    return [e.normalize() if isinstance(e, EllipticCurvePoint) else e for e in elements]


def _round_challenge(A, B, g_hat, k, Q, L_tilde):
    a, b, q = _norm(A, B, Q)
    return pivot.fiat_shamir_hash([a, b, g_hat, k, q, L_tilde], k.order)


def _halve(L_tilde, c, gf):
    assert L_tilde.constant == 0, "Next line assumes L_tilde is a linear form, not affine form."
    half = len(L_tilde) // 2
    left = pivot.LinearForm([coeff * gf(c) for coeff in L_tilde.coeffs[:half]])
    return left + pivot.LinearForm(L_tilde.coeffs[half:])


def protocol_4_prover(g_hat, k, Q, L_tilde, z_hat, gf, proof={}, round_i=0):
    while True:
        half = len(g_hat) // 2
        zero = [0] * half
        z_l, z_r = z_hat[:half], z_hat[half:]
        A = pivot.vector_commitment(z_l, int(L_tilde(zero + z_l)), g_hat[half:], k)
        B = pivot.vector_commitment(z_r, int(L_tilde(z_r + zero)), g_hat[:half], k)
        proof[f"A{round_i}"], proof[f"B{round_i}"] = A, B
        c = _round_challenge(A, B, g_hat, k, Q, L_tilde)
        g_hat = [(g_hat[i] ** c) * g_hat[half + i] for i in range(half)]
        Q = A * (Q ** c) * (B ** (c ** 2))
        L_tilde = _halve(L_tilde, c, gf)
        z_hat = [z_l[i] + c * z_r[i] for i in range(half)]
        if len(z_hat) <= 2:
            proof["z_prime"] = z_hat
            return proof
        round_i += 1


def _p5_public(generators, P, L, y, t, A, order):
    a, p = _norm(A, P)
    common = [t, a, generators, p, L, y]
    c0 = pivot.fiat_shamir_hash(common + [0] + [TAG], order)
    c1 = pivot.fiat_shamir_hash(common + [1] + [TAG], order)
    Q = A * (P ** c0) * (generators["k"] ** int(c1 * (c0 * y + t)))
    return c0, c1, Q, pivot.LinearForm(L.coeffs + [0]) * c1


def protocol_5_prover(generators, P, L, y, x, gamma, gf):
    n = len(x)
    L, y = pivot.affine_to_linear(L, y, n)
    assert bin(n + 1).count("1") == 1, \
        "This implementation requires n+1 to be power of 2 (else, use padding with zeros)."
    order = gf.order
    r = [prng.randrange(order) for _ in range(n)]
    rho = prng.randrange(order)
    t = L(r)
    A = pivot.vector_commitment(r, rho, generators["g"], generators["h"])
    proof = {"t": t, "A": A}
    c0, c1, Q, L_tilde = _p5_public(generators, P, L, y, t, A, order)
    z = [c0 * x[i] + r[i] for i in range(n)]
    z_hat = z + [gf(c0 * gamma + rho)]
    assert L(z) * c1 == L_tilde(z_hat)
    return protocol_4_prover(generators["g"] + [generators["h"]], generators["k"], Q, L_tilde, z_hat, gf, proof)


def protocol_4_verifier(g_hat, k, Q, L_tilde, gf, proof, round_i=0):
    while True:
        half = len(g_hat) // 2
        A, B = proof[f"A{round_i}"], proof[f"B{round_i}"]
        c = _round_challenge(A, B, g_hat, k, Q, L_tilde)
        g_hat = [(g_hat[i] ** c) * g_hat[half + i] for i in range(half)]
        Q = A * (Q ** c) * (B ** (c ** 2))
        L_tilde = _halve(L_tilde, c, gf)
        if len(g_hat) <= 2:
            z_prime = proof["z_prime"]
            return pivot.vector_commitment(z_prime, int(L_tilde(z_prime)), g_hat, k) == Q
        round_i += 1


def protocol_5_verifier(generators, P, L, y, proof, gf):
    L, y = pivot.affine_to_linear(L, y, len(generators["g"]))
    c0, c1, Q, L_tilde = _p5_public(generators, P, L, y, proof["t"], proof["A"], gf.order)
    return protocol_4_verifier(generators["g"] + [generators["h"]], generators["k"], Q, L_tilde, gf, proof)
