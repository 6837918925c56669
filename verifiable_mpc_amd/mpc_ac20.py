"""Counterpart of verifiable_mpc/ac20/mpc_ac20.py (the MPyC driver of the AC20 path) for ONE party of an
M-party proof over a SECRET-SHARED witness, with the party's group work on its MI355X.

    vector_commitment     mpc_ac20.py:35-42    secure_repeat(g + [h], x + [gamma])
    create_generators     mpc_ac20.py:45-51    kg = gather([secure_repeat(h, u) ...]); g = kg[1:], k = kg[0]
    protocol_4_prover     mpc_ac20.py:141-203  (fold of PUBLIC generators :176, z' opened :191)
    protocol_5_prover     mpc_ac20.py:206-269  (y, t, A opened :217-234)

Same coroutine names and positional arguments as the reference; what MPyC supplies there
(`mpyc.runtime.mpc`, `mpyc.secgroups.repeat_public_base_public_output`, secure field types) is supplied
here by a small stand-in, because MPyC exists in neither the build container nor the GPU box:

  * `SecureScalar` - this party's Shamir share of a value mod l, with the LINEAR arithmetic the path
    performs on secret values (`c0 * x_i + r_i`, `L(r)`, `z_l + c * z_r`: mpc_ac20.py:250,226,185);
  * `PartyRuntime` - `_random`, `output`, `gather` and the exchange of one group element per party;
  * `secure_repeat` - [mpyc-recall: repeat_public_base_public_output multiplies the party's shares by its
    Lagrange coefficient, does the multi-exponentiation LOCALLY, sends the resulting single group
    element to the others and multiplies the M elements].  The local step is the same (n+1)-term MSM as
    the single-party commitment with scalars lambda_p * share_p[i] - uniform 253-bit values - and runs
    on the same Pippenger kernel (csrc/msm.hip).

One party per GPU is plain replication: no GPU collective, the exchange of opened
points / scalars stays with the MPC runtime.  `LocalHub` plays that runtime for M parties inside one
process (M coroutines sharing one GPU) so that the party logic - Lagrange weights, what is opened
when, identical transcripts on every party - is testable without sockets; with M = 1 (threshold 0) the
run is what the reference's own Ed25519 tests execute (test/test_demo_zkp_mpc_ac20.py:17-23).

Limitation, stated plainly: there is no real MPyC on either side, so this is pinned against the
reference's mpc_ac20.py run over the build-written shim with m = 1 (tests/golden/mpc_ac20_m1.json)
and against the plain prover; a maintainer binds `PartyRuntime` to `mpyc.runtime.mpc` as
INTEGRATION.md shows.
"""
import asyncio

from . import compressed_pivot as cp
from . import pivot
from .device import PointVector, ScalarVector, reduce_scalar
from .fields import GF
from .groups import ORDER, Ed25519Point


# ---- secret-shared scalars -------------------------------------------------------------------------

class SecureScalar(pivot.SecureObject):
    """This party's share of a value mod l.  Only what is local in a linear secret-sharing scheme:
    sums of shares, and products with PUBLIC values (ints, field elements)."""
    __slots__ = ("share", "rt")

    def __init__(self, share, rt):
        self.share, self.rt = int(share) % ORDER, rt

    def _public(self, v):
        if isinstance(v, SecureScalar):
            return None
        return pivot._residue(v)

    def __add__(self, other):
        if isinstance(other, SecureScalar):
            return SecureScalar(self.share + other.share, self.rt)
        # a public constant is added by ONE party's share polynomial... for Shamir shares a constant
        # polynomial is every party's share of that constant
        return SecureScalar(self.share + self._public(other), self.rt)

    __radd__ = __add__

    def __neg__(self):
        return SecureScalar(-self.share, self.rt)

    def __sub__(self, other):
        return self + (-other if isinstance(other, SecureScalar) else -pivot._residue(other))

    def __rsub__(self, other):
        return (-self) + other

    def __mul__(self, other):
        if isinstance(other, SecureScalar):
            raise NotImplementedError("share x share needs a resharing round: not on the AC20 prover path")
        return SecureScalar(self.share * self._public(other), self.rt)

    __rmul__ = __mul__

    def __repr__(self):
        return "<secret share>"          # never enters a transcript


def _shares_of(values):
    """list of SecureScalar -> list of int shares (public values are shares of themselves)"""
    return [v.share if isinstance(v, SecureScalar) else pivot._residue(v) for v in values]


# ---- the slice of mpyc.runtime.mpc the path uses ---------------------------------------------------

class LocalHub:
    """In-process exchange between M party coroutines (the role of MPyC's TCP runtime)."""

    def __init__(self, parties):
        self.parties = parties
        self._slots = {}

    async def exchange(self, pid, tag, value):
        """every party contributes `value` under `tag`; returns the M values in party order"""
        slot = self._slots.get(tag)
        if slot is None:
            slot = self._slots[tag] = ([None] * self.parties, [0], asyncio.Event())
        values, count, done = slot
        values[pid] = value
        count[0] += 1
        if count[0] == self.parties:
            done.set()
        await done.wait()
        return list(values)


def recombination_vector(xs, x_r=0):
    """Lagrange coefficients for evaluation points xs at x_r, mod l (the role of
    mpyc.thresha._recombination_vector / verifiable_mpc/ac20/recombine.py:_recombination_vectors)."""
    out = []
    for i, x_i in enumerate(xs):
        num = den = 1
        for j, x_j in enumerate(xs):
            if i != j:
                num = num * (x_r - x_j) % ORDER
                den = den * (x_i - x_j) % ORDER
        out.append(num * pow(den, ORDER - 2, ORDER) % ORDER)
    return out


class PartyRuntime:
    """One party's view: pid in 0..M-1 holds the Shamir share at x = pid + 1 of a degree-`threshold`
    polynomial.  `rng` draws this party's randomness (`mpc._random`); `hub` connects the parties."""

    def __init__(self, pid=0, parties=1, threshold=0, rng=None, hub=None, gf=None):
        from random import SystemRandom
        assert 0 <= pid < parties and 0 <= threshold < parties
        self.pid, self.parties, self.threshold = pid, parties, threshold
        self.rng = rng or SystemRandom()
        self.hub = hub or LocalHub(parties)
        self.gf = gf or GF(ORDER)
        self.lagrange = recombination_vector(list(range(1, parties + 1)))[pid]
        self._tag = 0

    def _next_tag(self, kind):
        self._tag += 1              # every party runs the same program: the n-th exchange has the same tag
        return (kind, self._tag)

    def secret(self, share):
        return SecureScalar(share, self)

    def _random(self, sectype=None):
        """mpc._random(sectype): a uniformly random secret nobody knows (mpc_ac20.py:48,224-225).
        Every party draws its own share; M arbitrary shares are a sharing (of degree <= M - 1) of
        sum_p lambda_p * share_p, and since this module recombines with the M-point Lagrange vector
        and only ever combines shares LINEARLY, that is all the path needs.  [MPyC uses pseudo-random
        secret sharing of degree `threshold` here: same distribution of the secret.]"""
        return SecureScalar(self.rng.randrange(ORDER), self)

    async def output(self, x):
        """mpc.output: open secret scalar(s) to all parties (public values pass through)"""
        single = not isinstance(x, (list, tuple))
        vals = [x] if single else list(x)
        mine = [(v.share * self.lagrange % ORDER) if isinstance(v, SecureScalar) else None for v in vals]
        every = await self.hub.exchange(self.pid, self._next_tag("out"), mine)
        opened = []
        for j, v in enumerate(vals):
            if isinstance(v, SecureScalar):
                opened.append(self.gf(sum(part[j] for part in every) % ORDER))
            else:
                opened.append(v)
        return opened[0] if single else opened

    async def gather(self, *aws):
        if len(aws) == 1 and isinstance(aws[0], (list, tuple)):
            return list(await asyncio.gather(*aws[0]))
        return list(await asyncio.gather(*aws))

    async def exchange_points(self, points):
        """send this party's group elements, receive everybody's: list over parties of lists"""
        raw = [p.to_affine_bytes() for p in points]
        every = await self.hub.exchange(self.pid, self._next_tag("pts"), raw)
        return [[Ed25519Point.from_affine_bytes(b) for b in part] for part in every]


def deal(values, threshold, parties, rng):
    """Degree-`threshold` Shamir shares of each value for parties 1..M: shares[p][i] (dealer / test
    helper; in the reference the shares come from MPyC's input protocol)."""
    shares = [[] for _ in range(parties)]
    for v in values:
        coeffs = [int(v) % ORDER] + [rng.randrange(ORDER) for _ in range(threshold)]
        for p in range(parties):
            x = p + 1
            shares[p].append(sum(c * pow(x, k, ORDER) for k, c in enumerate(coeffs)) % ORDER)
    return shares


shamir_shares = deal          # round-1 name


# ---- secure_repeat: local MSM + one exchanged element -------------------------------------------------

def _product(points):
    acc = points[0]
    for p in points[1:]:
        acc = Ed25519Point.operation(acc, p)
    return acc.normalize()


def _local_msm(bases, shares, lam):
    """prod_i bases[i] ** (lam * share_i): this party's factor.  `bases`: list of points / PointVector
    (+ list tail); `shares`: int shares, the last len(tail) of them belong to the tail."""
    gv = pivot._points_on_device(bases)
    sv = shares if isinstance(shares, ScalarVector) else ScalarVector.from_ints(shares, gv.ctx)
    if lam != 1:
        sv = sv.scale(lam)                                                   # csrc/frvec.hip
    n = len(sv)
    # the last base / exponent pair rides along as the commitment's "h ** gamma" term
    return pivot._commit_launch(sv[:n - 1], pivot.DeviceScalar(_view_buf(sv, n - 1), gv.ctx), gv[:n - 1],
                                gv[n - 1], gv.ctx).result()


async def secure_repeat(a, x, rt=None):
    """mpyc.secgroups.repeat_public_base_public_output(a, x) for public base(s) `a` and secret
    exponent(s) `x`: the opened group element prod a_i ** x_i (mpc_ac20.py:41,49)."""
    bases = a if isinstance(a, (list, tuple, PointVector)) else [a]
    exps = x if isinstance(x, (list, tuple)) else [x]
    rt = rt or next(v.rt for v in exps if isinstance(v, SecureScalar))
    shares = _shares_of(exps)
    if len(bases) == 1:
        mine = Ed25519Point.repeat(cp._pt(bases[0]), shares[0] * rt.lagrange % ORDER)
    else:
        assert len(bases) >= len(shares), "Not enough generators."
        mine = _local_msm(bases, shares, rt.lagrange)
    every = await rt.exchange_points([mine])
    return _product([part[0] for part in every])


async def vector_commitment(x, gamma, g, h):
    """mpc_ac20.py:35-42."""
    if isinstance(g, PointVector):
        return await secure_repeat(g[:len(x)] + [cp._pt(h)], list(x) + [gamma])
    return await secure_repeat(list(g[:len(x)]) + [h], list(x) + [gamma])


async def create_generators(group, sectype, input_length, rt=None):
    """mpc_ac20.py:45-51: jointly random generators, nobody knows a discrete logarithm.  All
    input_length + 1 local exponentiations h ** (lambda * u_i) run as ONE fixed-base batch on the device
    (csrc/msm.hip k_fb_apply); one exchange carries all of them."""
    rt = rt or PartyRuntime()
    h = cp._pt(group.generator)
    random_exponents = [rt._random(sectype) for i in range(input_length + 1)]
    mine = PointVector.fixed_base(h, [u.share * rt.lagrange % ORDER for u in random_exponents], keep_proj=False)
    every = await rt.exchange_points(mine.to_points())
    kg = [_product([part[i] for part in every]) for i in range(input_length + 1)]
    return {"g": kg[1:], "h": h, "k": kg[0]}


# ---- Protocol 4 / 5, prover, over shares -----------------------------------------------------------------

async def protocol_4_prover(g_hat, k, Q, L_tilde, z_hat, gf, proof={}, round_i=0, rt=None, transcript=None):
    """mpc_ac20.py:141-203.  g_hat, k, Q, L_tilde are public; z_hat is a list of SecureScalar.  Each
    round: two local MSMs over this party's shares, one exchange carrying both partial points, the
    public fold of the generators on the device (:176), the local fold of the shares (:185)."""
    rt = rt or next(v.rt for v in z_hat if isinstance(v, SecureScalar))
    g_hat = pivot._points_on_device(g_hat)
    k = cp._pt(k)
    if not isinstance(transcript, cp._Transcript):
        transcript = cp._Transcript(transcript or "reference", k.order)
    lam = rt.lagrange
    while True:
        half = len(g_hat) // 2
        g_l, g_r = g_hat[:half], g_hat[half:]
        z_l, z_r = z_hat[:half], z_hat[half:]
        # exponents of k: L~(0 || z_l), L~(z_r || 0) - linear in the shares (:150-151)
        gamma_a = pivot._int(L_tilde([0] * half + z_l))
        gamma_b = pivot._int(L_tilde(z_r + [0] * half))
        sa = ScalarVector.from_ints(_shares_of(z_l) + _shares_of([gamma_a]), g_hat.ctx).scale(lam)
        sb = ScalarVector.from_ints(_shares_of(z_r) + _shares_of([gamma_b]), g_hat.ctx).scale(lam)
        A_loc, B_loc = pivot.vector_commitment_pair(sa[:half], pivot.DeviceScalar(_view_buf(sa, half), g_hat.ctx),
                                                    g_r, sb[:half],
                                                    pivot.DeviceScalar(_view_buf(sb, half), g_hat.ctx), g_l, k)
        every = await rt.exchange_points([A_loc, B_loc])
        A, B = _product([p[0] for p in every]), _product([p[1] for p in every])
        proof["A" + str(round_i)] = A
        proof["B" + str(round_i)] = B
        c = transcript.round_challenge(round_i, A, B, g_hat, k, Q, L_tilde)
        g_hat = g_l.fold(g_r, c)
        if transcript.mode == "reference":
            Q = cp._fold_commitment(A, Q, B, c)
        L_tilde = cp._fold_form(L_tilde, c, half, gf)
        z_hat = [z_l[i] + c * z_r[i] for i in range(half)]
        if len(z_hat) <= 2:
            proof["z_prime"] = await rt.output(z_hat)               # :187-191
            return proof
        round_i += 1


class _OffsetBuf:
    """a 32-byte window into a ScalarVector's buffer, shaped like a DeviceBuffer for DeviceScalar"""

    def __init__(self, ptr, keep):
        self.ptr, self._keep = ptr, keep


def _view_buf(sv, index):
    return _OffsetBuf(sv.ptr + 32 * index, sv)


async def protocol_5_prover(generators, P, L, y, x, gamma, gf, rt=None, transcript=None):
    """mpc_ac20.py:206-269.  x: list of SecureScalar (this party's shares of the witness), gamma:
    SecureScalar; y, L may carry secret parts and are opened first (:214-217), as are t and A (:228-230)."""
    mode = transcript or cp.TRANSCRIPT
    rt = rt or next(v.rt for v in x if isinstance(v, SecureScalar))
    g, h, k = generators["g"], cp._pt(generators["h"]), cp._pt(generators["k"])
    P = cp._pt(P)
    proof = {}
    n = len(x)
    L, y = pivot.affine_to_linear(L, y, n)
    L.constant = await rt.output(L.constant)
    y = await rt.output(y)
    assert bin(n + 1).count("1") == 1, \
        "This implementation requires n+1 to be power of 2 (else, use padding with zeros)."
    order = gf.order
    r = list(rt._random() for i in range(n))
    rho = rt._random()
    t = L(r)
    A = await vector_commitment(r, rho, g, h)
    t = await rt.output(t)
    proof["t"] = t
    proof["A"] = A
    gens_for_hash = {"g": g if isinstance(g, PointVector) or mode == "compact" else list(g),
                     "h": generators["h"], "k": generators["k"]}
    c0, c1, seed = cp._p5_challenges(mode, order, gens_for_hash, t, A, P, L, y)
    z = [c0 * x_i + r[i] for i, x_i in enumerate(x)]
    phi = c0 * gamma + rho
    z_hat = z + [phi]
    g_hat = pivot._points_on_device(g) + [h]
    Q = cp._LazyQ(A, P, k, c0, int(pivot._int(c1 * (c0 * y + t))), order)
    if mode == "reference":
        Q = Q.point()
    L_tilde = cp._extend_form(L, c1)
    return await protocol_4_prover(g_hat, k, Q, L_tilde, z_hat, gf, proof, rt=rt,
                                   transcript=cp._p5_setup(generators, k, seed, mode, order))


# ---- round-1 helpers kept for callers of the local step alone ---------------------------------------------

def local_commitment_share(x_shares, gamma_share, g, h, lagrange_coeff):
    """One party's factor of the opened commitment: prod_i g_i^(lambda*x_share_i) * h^(lambda*gamma_share)."""
    assert len(g) >= len(x_shares), "Not enough generators."
    lam = reduce_scalar(lagrange_coeff)
    gv = pivot._points_on_device(g)
    xs = pivot._scalars_on_device(x_shares).scale(lam)          # csrc/frvec.hip
    return pivot._commit_launch(xs, reduce_scalar(gamma_share) * lam % ORDER, gv, h, gv.ctx).result()


def combine_commitment_shares(points):
    """Product of the parties' elements (what every party computes after the exchange)."""
    return _product(list(points))


def install_mpc(reference_package="verifiable_mpc.ac20", runtime=None):
    """Point an importable reference at this module: rebinds the coroutines of
    verifiable_mpc.ac20.mpc_ac20 that sit on the hot path.  Needs a `PartyRuntime` bound to the real
    MPyC runtime for anything but a single in-process party."""
    import importlib
    ref = importlib.import_module(reference_package + ".mpc_ac20")
    patched = []
    for name in ("vector_commitment", "create_generators", "protocol_4_prover", "protocol_5_prover"):
        setattr(ref, name, globals()[name])
        patched.append(f"{ref.__name__}.{name}")
    return patched
