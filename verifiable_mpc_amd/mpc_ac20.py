"""Per-party local step of the MPC variant (verifiable_mpc/ac20/mpc_ac20.py:35-42).

In the reference's MPyC driver the Pedersen commitment of a SECRET-SHARED vector is
    c = secure_repeat(g + [h], x + [gamma])
[mpyc-recall: repeat_public_base_public_output - every party multiplies its Shamir shares by
its Lagrange coefficient, does the multi-exponentiation locally, sends the resulting single
group element to the others and all multiply the M elements].  The local step is the same
MSM as the single-party commitment, with scalars  lambda_p * share_p[i]  - uniform 253-bit
values - so it runs on the same kernel; with one party per GPU this is plain replication
(SURVEY.md 8e): no GPU collective, the exchange stays with the MPC runtime.

MPyC exists in neither the build container nor the GPU box, so only this local step and its
recombination are provided and tested (against the plain commitment); the asyncio driver,
`mpc.output`, `mpc.gather` stay with the reference.
"""
from .device import PointVector, ScalarVector, reduce_scalar
from .groups import ORDER, Ed25519Point
from . import pivot


def local_commitment_share(x_shares, gamma_share, g, h, lagrange_coeff):
    """One party's factor of the opened commitment: prod_i g_i^(lambda*x_share_i) * h^(lambda*gamma_share)."""
    assert len(g) >= len(x_shares), "Not enough generators."
    lam = reduce_scalar(lagrange_coeff)
    gv = pivot._points_on_device(g)
    xs = pivot._scalars_on_device(x_shares).scale(lam)          # csrc/frvec.hip
    return pivot._commit_launch(xs, reduce_scalar(gamma_share) * lam % ORDER, gv, h, gv.ctx).result()


def combine_commitment_shares(points):
    """Product of the parties' elements (what every party computes after the exchange)."""
    acc = Ed25519Point.identity
    for p in points:
        acc = Ed25519Point.operation(acc, p)
    return acc.normalize()


def recombination_vector(xs, x_r=0):
    """Lagrange coefficients for evaluation points xs at x_r, mod l
    (the role of verifiable_mpc/ac20/recombine.py:_recombination_vectors)."""
    out = []
    for i, x_i in enumerate(xs):
        num = den = 1
        for j, x_j in enumerate(xs):
            if i != j:
                num = num * (x_r - x_j) % ORDER
                den = den * (x_i - x_j) % ORDER
        out.append(num * pow(den, ORDER - 2, ORDER) % ORDER)
    return out


def shamir_shares(values, threshold, parties, rng):
    """Degree-`threshold` Shamir shares of each value for parties 1..M (test helper)."""
    shares = [[] for _ in range(parties)]
    for v in values:
        coeffs = [int(v) % ORDER] + [rng.randrange(ORDER) for _ in range(threshold)]
        for p in range(parties):
            x = p + 1
            shares[p].append(sum(c * pow(x, k, ORDER) for k, c in enumerate(coeffs)) % ORDER)
    return shares
