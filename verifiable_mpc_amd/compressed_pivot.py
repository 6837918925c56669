"""Drop-in counterpart of verifiable_mpc/ac20/compressed_pivot.py (AC20 Protocols 4 and 5)
with the O(N) work on MI355X.

    protocol_4_prover    compressed_pivot.py:29-86
    protocol_5_prover    compressed_pivot.py:89-145
    protocol_4_verifier  compressed_pivot.py:148-202
    protocol_5_verifier  compressed_pivot.py:205-239

Per halving round the GPU runs two Pippenger MSMs (A_i, B_i: csrc/msm.hip), the generator
fold g' (csrc/exact.hip), the scalar folds z', L' and the two inner products (csrc/frvec.hip)
and formats the Fiat-Shamir pre-image (csrc/format.hip); the host hashes it and does the
O(1) algebra on Q.

Two scalar-side modes, chosen by the argument types:
  * Python lists (the reference's calling convention): scalars stay Python objects with the
    reference's typing semantics (ints are never reduced, field elements are), so the
    pre-image text is what the reference would hash; only reduced copies go to the GPU;
  * device vectors (verifiable_mpc_amd.device.ScalarVector / PointVector): everything stays
    in HBM; this is the mode for N = 2^20.
Transcripts: "reference" = str(input_list) as in the reference (default);
"compact" = the build's O(1)-per-round byte transcript.
"""
import hashlib
import logging
import os
from random import SystemRandom

from . import _native, pivot
from .device import PointVector, ScalarVector, reduce_scalar
from .groups import EllipticCurvePoint as EllipticCurveElement
from .groups import Ed25519Point, as_point

prng = SystemRandom()

logger_cp = logging.getLogger("compressed_pivot")
logger_cp.setLevel(logging.INFO)
logger_cp_hin = logging.getLogger("compressed_pivot_hash_inputs")
logger_cp_hin.setLevel(logging.INFO)
logger_cp_hout = logging.getLogger("compressed_pivot_hash_outputs")
logger_cp_hout.setLevel(logging.INFO)

TRANSCRIPT = "reference"
CHUNK = 4096
# compact transcript, device mode: once the vector is this short the generator fold is pure
# latency (a 253-doubling chain per round); the remaining rounds keep the generators fixed and
# put the challenge products into the MSM scalars instead (csrc/frvec.hip k_fr_tail_scalars)
TAIL_BASE = 1 << 16
# reference transcript, device mode: down to this vector length the NEXT round's A, B are committed over the vector as it
# is BEFORE its fold (the challenge goes into the scalars), beside the fold instead of behind it (_early_pair_*).
# OFF (0): measured in rounds 4 and 5 (EXPERIMENTS.md R5.9) it LOSES - 485 against 472 ms per proof: the pair no longer
# waits for the fold (15 instead of 42 ms inside vector_commitment_pair), but its four commitments take the vector
# ALUs the fold needs, the text arrives later and the hash waits longer than before.  VMPC_EXPERIMENTAL=1
# VMPC_EARLY_PAIR_MIN=16384 turns it on; parity is the same either way (tests/test_gpu_protocol.py runs both).
EARLY_PAIR_MIN = int(os.environ.get("VMPC_EARLY_PAIR_MIN", "0")) if os.environ.get("VMPC_EXPERIMENTAL", "0") != "0" else 0
EARLY_PAIR_MAX = int(os.environ.get("VMPC_EARLY_PAIR_MAX", str(1 << 62)))   # ... and only up to this length
EARLY_PAIR_FIRST = os.environ.get("VMPC_EARLY_PAIR_FIRST", "1") != "0"     # the pair ahead of the fold, not beside it
# reference transcript, device mode, tabulated CRS (round 6): while the vector is at least this long, a round's A_i, B_i
# come from the ROUND CONTEXT (csrc/prover.hip: commitments over the UNFOLDED, tabulated generators with the challenge
# products in the scalars - the same group elements) BEFORE the round's exact generator fold is enqueued.  The fold
# (8.6 ms of arithmetic at 2^19 elements) is still needed - its (X:Y:Z) are the bulk of the next pre-image - but the hash
# no longer waits for it AND the pair over the folded vector: 11.7 -> ~2.5 ms before round 1's hash, 6.8 -> 2.5 before
# round 2's.  Below the threshold a pass over the full table (1.3 ms whatever the round) costs more than the short fold
# + pair it replaces.  0: never.
REF_TABLE_PAIR_MIN = int(os.environ.get("VMPC_REF_TABLE_PAIR_MIN", "2"))
# ... and ALL the way down (round 6, second step; the first one stopped at 2^17 elements): the context folds ITS generators
# once, after REF_TABLE_JUMP_K challenges, in one pass over the table (csrc/fold_jump.hip; 2.9 ms into 2^15 columns,
# twice that into 2^16) - a LAZY fold (vmpc_p4_create_opts): the round that feeds the last of them still commits over the
# unfolded table, the fold is asked for once that pair has been collected (vmpc_p4_prefold) and runs on the GPU while the
# host hashes the next round's text - and every later pair is a commitment over the folded table (the
# fused short path, 0.25 ms) instead of an MSM over the exactly folded short vector (0.55 ms behind a 0.78-ms fold).
# Measured (scripts/ref_stall_probe.py, time outside sha256.update): 44.7 / 41.1 / 40.0 / 41.7 ms at 3 / 4 / 5 / 6
# challenges against 41.4 with the context closed at 2^17 elements.
REF_TABLE_JUMP_K = int(os.environ.get("VMPC_REF_TABLE_JUMP_K", "5"))
# the context on a stream of its own: a round's pair is asked for right after the FIRST slice of the round's exact fold
# (a short one, device.PointVector.fold after_first) is enqueued on the main stream and runs beside it; the other
# slices follow the pair.  Small rounds: 0.57 ms of gap + 0.48 of waiting for text instead of 0.42 + 0.83; big rounds
# 3.0 + 0.05 instead of 2.0 + 1.1; 34.3 against 38.0 ms outside the hash (profiles/r06_ref_pair_side.txt).  (With ALL
# the fold's slices enqueued ahead of the pair it lost, 450.5 against 444.4 ms: the fold's kernels starve the pair.)
REF_TABLE_PAIR_SIDE_STREAM = os.environ.get("VMPC_REF_TABLE_PAIR_SIDE", "1") != "0"


# ---- group glue on single elements (independent of the is_additive/is_multiplicative flags) ----

_pt = as_point      # our points, or any 3-coordinate projective element (e.g. an MPyC point) converted


def _gmul(a, b):
    return Ed25519Point.operation(a, b)


def _gpow(a, n):
    return Ed25519Point.repeat(a, int(n))


def _same_residue(a, b, order):
    return (pivot._residue(a) - pivot._residue(b)) % order == 0


class _GroupCheck:
    """The order-l ladder of _valid_group_elements in flight on a side stream; result() joins it."""

    def __init__(self, verdict=None, ctx=None, out=None, raw=None, keepalive=None):
        self.verdict, self.ctx, self.out, self.raw, self.keepalive = verdict, ctx, out, raw, keepalive

    def result(self):
        if self.verdict is None:
            from .groups import P as FIELD_P
            raw, n = self.raw, len(self.raw) // 64
            got = self.ctx.download(self.out.ptr, 64 * n).tobytes()        # synchronises the side stream
            self.verdict = True
            for i in range(n):
                x = int.from_bytes(raw[64 * i:64 * i + 32], "little")
                want = ((FIELD_P - x) % FIELD_P).to_bytes(32, "little") + raw[64 * i + 32:64 * i + 64]
                if got[64 * i:64 * i + 64] != want:
                    self.verdict = False
                    break
            self.keepalive = self.out = None
        return self.verdict


def _valid_group_elements_begin(points):
    """_valid_group_elements with the 253-step ladder (0.9 ms of single-lane latency for the 40 points of a
    2^20 proof) left running on a side stream: the caller goes on and joins the verdict before it accepts."""
    import numpy as np
    from .device import get_aux_context
    from .groups import ORDER
    try:
        pts = [_pt(p) for p in points]
        raw = b"".join(p.to_affine_bytes() for p in pts)
    except Exception:
        return _GroupCheck(False)
    n = len(pts)
    if n == 0:
        return _GroupCheck(True)
    ctx = get_aux_context(3)
    buf = ctx.upload(np.frombuffer(raw, np.uint8))
    if ctx.validate_points(buf.ptr, n):
        return _GroupCheck(False)
    sc = ctx.upload(np.frombuffer((ORDER - 1).to_bytes(32, "little") * n, np.uint8))
    out = ctx.alloc(64 * n)
    ctx.repeat(buf.ptr, n, True, sc.ptr, n, False, None, out.ptr)
    return _GroupCheck(None, ctx, out, raw, (buf, sc))


def _valid_group_elements(points, ctx=None):
    """True iff every point is a canonical element of the order-l subgroup of Ed25519.

    The verifier's inputs from the prover (P, A, A_i, B_i) are untrusted.  The kernels assume curve
    points (the niels mixed addition absorbs (0, 0)) and reduce exponents mod l, which equals the
    reference's unreduced `B ** (c ** 2)` (compressed_pivot.py:66,180) only on the order-l subgroup -
    so anything else is rejected: on-curve + canonical coordinates (vmpc_points_validate_dev), then
    (l - 1) * X == -X for all of them in one launch of the ladder kernel (vmpc_repeat_dev)."""
    return _valid_group_elements_begin(points).result()


def _proof_points(proof):
    """A, A_0.., B_0.. of a Protocol-4/5 proof dict (whatever is present)"""
    return [v for key, v in proof.items()
            if key == "A" or (key[:1] in "AB" and key[1:].isdigit())]


# ---- scalar-side strategies --------------------------------------------------------------------

def _on_device(*vs):
    return any(isinstance(v, ScalarVector) for v in vs)


def _coeffs_dev(form):
    return pivot._as_device(form.coeffs)


def _commit(z_part, gamma, g_part, k):
    return pivot.vector_commitment(z_part, gamma, g_part, k)


# ---- compact transcript ---------------------------------------------------------------------------

def _chunked_digest(tag, data):
    mv = memoryview(data)
    leaves = b"".join(hashlib.sha256(mv[o:o + CHUNK]).digest() for o in range(0, len(mv), CHUNK))
    return hashlib.sha256(tag + len(mv).to_bytes(8, "little") + leaves).digest()


def _chunked_digest_dev(tag, ctx, ptr, nbytes):
    """same digest with the leaves hashed on the device (csrc/sha256.hip)"""
    leaves = ctx.sha256_chunks(ptr, nbytes, CHUNK)
    return hashlib.sha256(tag + int(nbytes).to_bytes(8, "little") + leaves).digest()


def _sc_bytes(v):
    return reduce_scalar(v).to_bytes(32, "little")


def generators_digest(generators):
    """Digest of the CRS (g, h, k) for the compact transcript.  Cached on the device vector g, keyed
    by the h and k it was computed with: the same g under another h or k is another CRS."""
    g = generators["g"]
    h, k = _pt(generators["h"]), _pt(generators["k"])
    key = (h.to_affine_bytes(), k.to_affine_bytes())
    if isinstance(g, PointVector) and g._digest is not None and key in g._digest:
        return g._digest[key]
    gv = pivot._points_on_device(g)
    full = PointVector(gv.a, None, gv.ctx).concat([h, k])
    d = _chunked_digest_dev(b"vmpc-ac20/gens/v1", full.ctx, full.affine_ptr, 64 * len(full))
    if isinstance(g, PointVector):
        if g._digest is None:
            g._digest = {}
        g._digest[key] = d
    return d


def _form_digest_begin(L):
    """start hashing the form's coefficients on a side stream (the prover's announcement MSM runs meanwhile)"""
    c = L.coeffs
    if isinstance(c, ScalarVector) and len(c):
        from .device import get_aux_context
        side = get_aux_context(2)
        side.wait_for(c.ctx)
        L._pending_leaves = side.sha256_chunks_begin(c.ptr, 32 * len(c), CHUNK, keepalive=c)


def _form_digest(L):
    c = L.coeffs
    done = getattr(L, "_form_digest", None)
    if done is not None:
        L._form_digest = None              # one use: the coefficients may change afterwards
        return done
    pending = getattr(L, "_pending_leaves", None)
    if pending is not None:
        L._pending_leaves = None
        return hashlib.sha256(b"vmpc-ac20/form/v1" + (32 * len(c)).to_bytes(8, "little") + pending.result()).digest()
    if isinstance(c, ScalarVector):
        return _chunked_digest_dev(b"vmpc-ac20/form/v1", c.ctx, c.ptr, 32 * len(c))
    else:
        data = b"".join(_sc_bytes(pivot._residue(v)) for v in c)
    return _chunked_digest(b"vmpc-ac20/form/v1", data)


def _compact_seed(generators, P, L, y, t, A):
    return hashlib.sha256(b"vmpc-ac20/p5/v1" + generators_digest(generators) + _form_digest(L)
                          + P.to_affine_bytes() + _sc_bytes(pivot._residue(y))
                          + _sc_bytes(pivot._residue(t)) + A.to_affine_bytes()).digest()


def _challenge(digest, order):
    return int.from_bytes(digest, "little") % order


class _Transcript:
    """Challenge source for Protocol 4: reference text hashing or the compact chain."""

    def __init__(self, mode, order, state=None):
        if mode not in ("reference", "compact"):
            raise ValueError(f"unknown transcript mode {mode!r}")
        self.mode, self.order, self.state = mode, order, state

    def round_challenge(self, round_i, A, B, g_hat, k, Q, L_tilde, who="protocol_4_prover"):
        if self.mode == "reference":
            # compressed_pivot.py:51-59: A, B, Q normalised; g_hat, k as they are
            input_list = [A.normalize(), B.normalize(), g_hat, k, Q.normalize(), L_tilde]
            pivot.log_hash_input(logger_cp_hin, who, input_list)
            return pivot.fiat_shamir_hash(input_list, self.order)
        self.state = hashlib.sha256(self.state + round_i.to_bytes(4, "little")
                                    + A.to_affine_bytes() + B.to_affine_bytes()).digest()
        return _challenge(self.state, self.order)


# ---- Protocol 4 --------------------------------------------------------------------------------------
WIDE_TABLE_ROWS = 13      # device.PointVector.precompute(rows=13): csrc/msm.hip msm_table_batch_wide

def _tabulated(g_hat, k, whole=False):
    """g_hat and k live in one fixed-base table: commitments over the UNFOLDED g_hat are then cheaper
    than folding it (a 2^19-element fold costs as much as eight 2^20-term table commitments).
    whole: g_hat must be ALL of the table's generators (+ its leading extras), not a strict prefix - the
    round context (vmpc_p4_create) derives N from the table, a prefix has to take the round-by-round path."""
    t = getattr(g_hat, "_table", None)
    if t is None or t.rows == WIDE_TABLE_ROWS:
        # (a wide-window table serves commitments only: the fold jump and the bucket-free short rounds read rows spaced
        # 256 / rows bits)
        return False
    if whole and len(g_hat) - g_hat._table_tail != t.n:
        return False
    slot = t.extra_index(k)
    return slot is not None and slot >= g_hat._table_tail


def _round_prover_scalars(L_tilde, z_hat, half, gf, exponents=True):
    """Exponents of k in A_i, B_i and the split witness (compressed_pivot.py:35-42).
    exponents=False: the split only (the round's A_i, B_i are on their way already)."""
    if _on_device(L_tilde.coeffs, z_hat):
        Lc, z = _coeffs_dev(L_tilde), pivot._as_device(z_hat)
        z_l, z_r = z[:half], z[half:]
        if not exponents:
            return z_l, z_r, None, None
        # the two inner products stay on the device: they are only ever exponents of k in A_i, B_i
        gamma_a = Lc[half:].dot_dev(z_l)       # L~(0 || z_l)
        gamma_b = Lc[:half].dot_dev(z_r)       # L~(z_r || 0)
        return z_l, z_r, gamma_a, gamma_b
    z_l, z_r = z_hat[:half], z_hat[half:]
    gamma_a = int(L_tilde([0] * half + z_l))
    gamma_b = int(L_tilde(z_r + [0] * half))
    return z_l, z_r, gamma_a, gamma_b


def _fold_form(L_tilde, c, half, gf):
    """L' = c * L_l + L_r (compressed_pivot.py:70-73)."""
    assert L_tilde.constant == 0, "Next line assumes L_tilde is a linear form, not affine form."
    if isinstance(L_tilde.coeffs, ScalarVector):
        Lc = L_tilde.coeffs
        return pivot.AffineForm(Lc[:half].axpy(c, Lc[half:]), 0)
    scaled = [coeff * gf(c) for coeff in L_tilde.coeffs[:half]]
    return pivot.LinearForm(scaled) + pivot.LinearForm(L_tilde.coeffs[half:])


def _fold_witness(z_l, z_r, c, half):
    """z' = z_l + c * z_r (compressed_pivot.py:76)."""
    if isinstance(z_l, ScalarVector):
        return z_r.axpy(c, z_l)
    return [z_l[i] + c * z_r[i] for i in range(half)]


class _LazyPoint:
    """A commitment still being computed on a side stream; behaves like the point for what
    Protocol 4 does with Q (normalize() for the hash, use as an MSM input)."""

    def __init__(self, pending):
        self._pending, self._pt = pending, None

    def resolve(self):
        if self._pt is None:
            self._pt = self._pending.result()
            self._pending = None
        return self._pt

    def normalize(self):
        return self.resolve().normalize()

    def __eq__(self, other):
        return self.resolve() == (other.resolve() if isinstance(other, _LazyPoint) else other)

    def __repr__(self):
        return repr(self.resolve())


def _fold_commitment(A, Q, B, c, order=None):
    """Q' = A * Q**c * B**(c**2) (compressed_pivot.py:66; the exponent c**2 is not reduced
    in the reference, which changes nothing for an element of order l).  On the host, in C
    (vmpc_ed25519_fold_commitment_host: ~0.1 ms): only the normalised value enters the next hash, and as a
    3-term MSM on a side stream the product queued behind the round's generator fold."""
    order = order or Ed25519Point.order
    if isinstance(Q, _LazyPoint):
        Q = Q.resolve()
    raw = _native.fold_commitment_host(A.to_affine_bytes(), Q.to_affine_bytes(), B.to_affine_bytes(), c % order)
    return Ed25519Point.from_affine_bytes(raw)


def _unfold_commitment(Q0, rounds, order, ctx=None, wait=True):
    """Q_R from Q' = A * Q**c * B**(c**2) applied R times, as ONE (2R+1)-term MSM:
    Q_R = (prod_j c_j) Q_0 + sum_i (prod_{j>i} c_j) (A_i + c_i^2 B_i).
    wait=False: the launched commitment (its .result() is the point)."""
    scalars, points = [], []
    suffix = 1
    for A, B, c in reversed(rounds):
        scalars += [suffix, suffix * c * c % order]
        points += [A, B]
        suffix = suffix * c % order
    if isinstance(Q0, _LazyPoint):
        Q0 = Q0.resolve()
    q_terms = Q0.terms if isinstance(Q0, _LazyQ) else [(1, Q0)]
    for sc, pt in q_terms:
        scalars.append(suffix * sc % order)
        points.append(pt)
    pv = PointVector.from_points(points, ctx, keep_proj=False)
    pending = pivot._commit_launch(ScalarVector.from_ints(scalars, pv.ctx), 0, pv, Ed25519Point.identity, pv.ctx)
    return pending.result() if wait else pending


NATIVE_ROUNDS = os.environ.get("VMPC_NATIVE_ROUNDS", "1") != "0"
NATIVE_CHAIN = os.environ.get("VMPC_NATIVE_CHAIN", "1") != "0"      # the compact challenge chain inside the C call
# the pairs of the rounds before the fold jump over the CRS's wide-window table when it holds one (A/B knob)
USE_WIDE_COMMIT_TABLE = os.environ.get("VMPC_P4_WIDE_TABLE", "1") != "0"


def _protocol_4_native_rounds(g_hat, k, L_tilde, z_hat, gf, proof, round_i, transcript):
    """All halving rounds through the device-resident round context (csrc/prover.hip, vmpc_p4_*): one C call
    per round, only A_i, B_i (2 x 64 bytes) out and the challenge in; the compact hash chain stays here."""
    from ._native import P4Rounds
    table = g_hat._table
    assert L_tilde.constant == 0, "Next line assumes L_tilde is a linear form, not affine form."
    Lc = _coeffs_dev(L_tilde)
    def run():
        rounds = P4Rounds(g_hat.ctx, table, g_hat._table_tail, table.extra_index(k), z_hat.ptr, Lc.ptr,
                          n_total=len(z_hat), commit_table=USE_WIDE_COMMIT_TABLE and getattr(g_hat, "_wide", None) or None)
        try:
            n_rounds = len(z_hat).bit_length() - 2
            if NATIVE_CHAIN:
                state, pairs, z_prime = rounds.run_compact(transcript.state, round_i, n_rounds)
                transcript.state = state
                for i, (a, b) in enumerate(pairs):
                    proof["A" + str(round_i + i)] = Ed25519Point.from_affine_bytes(a)
                    proof["B" + str(round_i + i)] = Ed25519Point.from_affine_bytes(b)
            else:
                c = None
                for i in range(n_rounds):
                    a, b = rounds.round(c)
                    A, B = Ed25519Point.from_affine_bytes(a), Ed25519Point.from_affine_bytes(b)
                    proof["A" + str(round_i + i)] = A
                    proof["B" + str(round_i + i)] = B
                    c = transcript.round_challenge(round_i + i, A, B, None, k, None, None)
                z_prime = rounds.finish(c)
            proof["z_prime"] = [gf(v) for v in z_prime]
        finally:
            rounds.close()
    state0 = transcript.state
    try:
        run()
    except _native.VmpcError as e:
        # a round's commitments over the folded vector's table took the fused short path and met scalars beyond its
        # capacities (csrc/msm_short.hip: VMPC_E_AGAIN, reported when the rounds finish): the whole of Protocol 4 once
        # more on the general path - its inputs (z_hat, L~) were copied into the round context, they are intact
        if e.code != _native.E_AGAIN:
            raise
        transcript.state = state0
        g_hat.ctx.on_general_path(run)
    return proof


def protocol_4_prover(g_hat, k, Q, L_tilde, z_hat, gf, proof={}, round_i=0, transcript=None):
    """Non-interactive Protocol 4, prover (compressed_pivot.py:29-86); the reference's
    recursion is a loop here, `round_i` keeps its meaning."""
    g_hat = pivot._points_on_device(g_hat)
    k = _pt(k)
    Q = Q if isinstance(Q, (_LazyQ, _LazyPoint)) else _pt(Q)
    if not isinstance(transcript, _Transcript):
        transcript = _Transcript(transcript or "reference", k.order)
    if transcript.mode == "reference" and isinstance(Q, _LazyQ):
        Q = Q.point()
    tail_cs = None           # challenges not yet applied to g_hat (compact tail)
    early = None             # this round's A, B, launched during the previous round (_early_pair)
    return _protocol_4_prover_loop(g_hat, k, Q, L_tilde, z_hat, gf, proof, round_i, transcript, tail_cs, early)


def _ref_table_rounds(g_hat, k, L_tilde, z_hat, transcript):
    """the round context for the reference-transcript prover's big rounds, or None (see REF_TABLE_PAIR_MIN)"""
    m = len(z_hat) if hasattr(z_hat, "__len__") else 0
    if not (REF_TABLE_PAIR_MIN and NATIVE_ROUNDS and transcript.mode == "reference" and isinstance(z_hat, ScalarVector)
            and isinstance(L_tilde.coeffs, ScalarVector) and len(g_hat) == m and m >= max(8, 2 * REF_TABLE_PAIR_MIN)
            and m & (m - 1) == 0 and L_tilde.constant == 0 and _tabulated(g_hat, k, whole=True)):
        return None
    from ._native import P4Rounds
    from .device import get_aux_context
    table = g_hat._table
    # on a stream of its own: a round's pair then runs BESIDE the first slice of the exact fold that is enqueued on the
    # main stream just before it (a 2^14-element slice is a 1-ms ladder on a sixteenth of the chip's lanes) instead of in
    # front of it, and the fold's first text has landed by the time the pair is back
    side = get_aux_context(7) if REF_TABLE_PAIR_SIDE_STREAM else g_hat.ctx
    if side is not g_hat.ctx:
        side.wait_for(g_hat.ctx)                 # z_hat and L~ were produced on the main stream
    return P4Rounds(side, table, g_hat._table_tail, table.extra_index(k), z_hat.ptr, _coeffs_dev(L_tilde).ptr,
                    n_total=m, commit_table=USE_WIDE_COMMIT_TABLE and getattr(g_hat, "_wide", None) or None,
                    jump_k=REF_TABLE_JUMP_K, lazy_fold=True)


def _protocol_4_prover_loop(g_hat, k, Q, L_tilde, z_hat, gf, proof, round_i, transcript, tail_cs, early):
    if _on_device(L_tilde.coeffs, z_hat):
        z_hat = pivot._as_device(z_hat)
    # reference transcript: the round context that supplies the rounds' pairs; released with this call whatever way it
    # ends (the context's arena goes back to the vmpc_ctx's pool: the next proof must not find it busy)
    table_rounds = _ref_table_rounds(g_hat, k, L_tilde, z_hat, transcript)
    try:
        return _protocol_4_prover_rounds(g_hat, k, Q, L_tilde, z_hat, gf, proof, round_i, transcript, tail_cs, early,
                                         table_rounds)
    finally:
        if table_rounds is not None:
            table_rounds.close()


def _protocol_4_prover_rounds(g_hat, k, Q, L_tilde, z_hat, gf, proof, round_i, transcript, tail_cs, early, table_rounds):
    fed = 0                  # challenges the context has been given since its last fold
    if table_rounds is not None:
        a0, b0 = table_rounds.round(None)
        first = (Ed25519Point.from_affine_bytes(a0), Ed25519Point.from_affine_bytes(b0))
        early = lambda: first                                                    # noqa: E731
    while True:
        if _on_device(L_tilde.coeffs, z_hat):
            z_hat = pivot._as_device(z_hat)
        m = len(z_hat)
        half = m // 2
        if tail_cs is None and NATIVE_ROUNDS and transcript.mode == "compact" and isinstance(z_hat, ScalarVector) \
                and isinstance(L_tilde.coeffs, ScalarVector) and len(g_hat) == m and m >= 4 and m & (m - 1) == 0 \
                and _tabulated(g_hat, k, whole=True):
            return _protocol_4_native_rounds(g_hat, k, L_tilde, z_hat, gf, proof, round_i, transcript)
        z_l, z_r, gamma_a, gamma_b = _round_prover_scalars(L_tilde, z_hat, half, gf, exponents=early is None)
        logger_cp.debug("Calculate A_i, B_i.")
        if tail_cs is None and transcript.mode == "compact" and isinstance(z_l, ScalarVector) \
                and len(g_hat) == m and m >= 4 and (len(g_hat) <= TAIL_BASE or _tabulated(g_hat, k)):
            tail_cs = []
            tail_products = ScalarVector.empty(len(g_hat), g_hat.ctx)     # challenge products per generator
        if early is not None:
            A, B = early()
            early = None
        elif tail_cs is not None:
            ctx = g_hat.ctx
            v_a, v_b = ScalarVector.empty(len(g_hat), ctx), ScalarVector.empty(len(g_hat), ctx)
            ctx.fr_tail_scalars_inc(tail_cs[-1] if tail_cs else 0, len(tail_cs), len(g_hat).bit_length() - 1,
                                    z_hat.ptr, tail_products.ptr, v_a.ptr, v_b.ptr)
            A, B = pivot.vector_commitment_pair(v_a, gamma_a, g_hat, v_b, gamma_b, g_hat, k)
        else:
            g_l, g_r = g_hat[:half], g_hat[half:]
            A, B = pivot.vector_commitment_pair(z_l, gamma_a, g_r, z_r, gamma_b, g_l, k)
        proof["A" + str(round_i)] = A
        proof["B" + str(round_i)] = B

        c = transcript.round_challenge(round_i, A, B, g_hat, k, Q, L_tilde)
        logger_cp_hout.debug(f"After hash, hash=\n{c}")

        L_next = _fold_form(L_tilde, c, half, gf)
        z_next = _fold_witness(z_l, z_r, c, half)
        if tail_cs is not None:
            tail_cs.append(c)
        else:
            g_l, g_r = g_hat[:half], g_hat[half:]
            unfolded = g_hat
            ahead = transcript.mode == "reference" and isinstance(z_next, ScalarVector) and len(g_hat) == m \
                and isinstance(L_next.coeffs, ScalarVector) and 0 < EARLY_PAIR_MIN <= m <= EARLY_PAIR_MAX \
                and m & (m - 1) == 0 and table_rounds is None
            # (its scalars on the main stream and two side streams ordered behind them BEFORE the fold is enqueued)
            prep = _early_pair_prepare(g_hat.ctx, L_next, z_next, c, half, gf) if ahead else None
            pair_after_fold = False
            if table_rounds is not None:
                if half >= 2 * REF_TABLE_PAIR_MIN:
                    fed += 1
                    if table_rounds.ctx is not g_hat.ctx:
                        pair_after_fold = True          # (own stream: asked for right after the fold is enqueued)
                    else:
                        # the NEXT round's pair from the round context, before this round's exact fold is enqueued
                        an, bn = table_rounds.round(c)
                        nxt = (Ed25519Point.from_affine_bytes(an), Ed25519Point.from_affine_bytes(bn))
                        early = lambda nxt=nxt: nxt                              # noqa: E731
                else:
                    table_rounds.close()
                    table_rounds = None
            # reference transcript: the folded vector's text is the bulk of the next pre-image - folded, formatted
            # and copied slice by slice (PointVector.fold), hashed while the rest is still on its way
            if ahead and EARLY_PAIR_FIRST:
                # the pair FIRST, the fold ordered behind it (round 6): beside a fold that fills the vector ALUs for
                # 8.6 ms the four commitments took ~40 ms to come back (EXPERIMENTS R5.9); ahead of it they take ~2 ms,
                # and the fold's first slice of text follows ~1 ms later
                early = _early_pair_launch(unfolded, k, half, prep, then=g_hat.ctx)
            fold_due = table_rounds is not None and fed == REF_TABLE_JUMP_K
            if fold_due:
                fed = -(1 << 30)
            hook = None
            if pair_after_fold:
                # own stream: the pair is ENQUEUED right after the exact fold's first slice (it runs beside it; the
                # other slices are ordered behind it) and collected when the next round asks for it - the host work
                # in between (the rest of the fold's launches, Q, L~'s text) no longer waits for the pair
                def hook(more, rounds=table_rounds, main=g_hat.ctx):
                    rounds.round_begin(c)
                    if more:
                        main.wait_for(rounds.ctx)

                def early(rounds=table_rounds, fold_due=fold_due):
                    an, bn = rounds.round_end()
                    if fold_due:
                        # the context's one fold of its generators, behind the pair, under the next hash
                        rounds.prefold()
                    return Ed25519Point.from_affine_bytes(an), Ed25519Point.from_affine_bytes(bn)
            g_hat = g_l.fold(g_r, c, stream_text=transcript.mode == "reference", after_first=hook)
            if fold_due and not pair_after_fold:
                # the context's one fold of its generators, behind the exact fold just enqueued, under the next hash
                table_rounds.prefold()
            if ahead and not EARLY_PAIR_FIRST:
                early = _early_pair_launch(unfolded, k, half, prep)
        if transcript.mode == "reference":
            # only the reference pre-image contains Q (compressed_pivot.py:52); the compact
            # chain binds Q once at the start, so the prover need not track it
            Q = _fold_commitment(A, Q, B, c)
        L_tilde = L_next
        if transcript.mode == "reference" and isinstance(L_tilde.coeffs, ScalarVector):
            L_tilde.coeffs.text_begin()
        z_hat = z_next
        if len(z_hat) <= 2:
            if isinstance(z_hat, ScalarVector):
                z_hat = [gf(v) for v in z_hat.to_ints()]
            proof["z_prime"] = z_hat
            return proof
        round_i += 1


def _early_pair_prepare(main, L_next, z_next, c, half, gf):
    """The next round's A, B WITHOUT waiting for this round's generator fold (reference transcript, device vectors).

    The reference's next round commits to halves of the folded vector g' = [(g_l[i] ** c) * g_r[i]]
    (compressed_pivot.py:41-42,64); an exact 2^19-element fold takes 8.6 ms and the pair leads the next pre-image
    (compressed_pivot.py:52), so hashing cannot start before it.  With q = half / 2 and z' = z'_l || z'_r:
        A' = <z'_l, g'_r> = <c z'_l, g_l[q:]> + <z'_l, g_r[q:]>        B' = <z'_r, g'_l> = <c z'_r, g_l[:q]> + <z'_r, g_r[:q]>
    - four commitments over slices of the vector as it is NOW, the same group elements (twice the terms: 2 ms of
    work at 2^20 against the fold's 8.6 on the critical path).  The fold still runs - its text is the bulk of that
    pre-image - but beside the pair, slice by slice under the hash, not in front of it.
    This half: the scalars (on the main stream) and two side streams ordered behind THEM - called before the fold is
    enqueued on the main stream, so that what runs on the side streams does not wait for it."""
    from .device import get_aux_context
    q = half // 2
    z_l, z_r, gamma_a, gamma_b = _round_prover_scalars(L_next, z_next, q, gf)
    cz_l, cz_r = z_l.scale(c), z_r.scale(c)
    sa, sb = get_aux_context(5), get_aux_context(6)
    sa.wait_for(main)
    sb.wait_for(main)
    return q, (z_l, z_r, cz_l, cz_r, gamma_a, gamma_b), (sa, sb)


def _early_pair_launch(g_hat, k, half, prep, then=None):
    """... and this half: the four commitments over the unfolded vector on the side streams -> a callable that
    collects (A', B').  then: a context whose LATER work (the fold) is ordered behind the four commitments."""
    q, (z_l, z_r, cz_l, cz_r, gamma_a, gamma_b), (sa, sb) = prep
    g_l, g_r = g_hat[:half], g_hat[half:]
    parts = [pivot._commit_launch(cz_l, 0, g_l[q:], k, sa), pivot._commit_launch(z_l, gamma_a, g_r[q:], k, sa),
             pivot._commit_launch(cz_r, 0, g_l[:q], k, sb), pivot._commit_launch(z_r, gamma_b, g_r[:q], k, sb)]
    if then is not None:
        then.wait_for(sa)
        then.wait_for(sb)

    def collect():
        pts = [p.result() for p in parts]
        return _gmul(pts[0], pts[1]), _gmul(pts[2], pts[3])
    return collect


def _protocol_4_verifier_compact(g_hat, k, Q, L_tilde, gf, proof, round_i, transcript):
    """Compact-transcript verifier without materialising any folded vector.

    The challenges only depend on (state, A_i, B_i), so they are all known up front; R folds of
    g_hat and L_tilde (compressed_pivot.py:178,185-188) are one linear map with coefficients
    s[j] = prod_i (c_i if the i-th index bit from the top is 0), hence the final check
    (:193-197)  z'_0 g'_0 + z'_1 g'_1 + L'(z') k == Q_R  is ONE N-term MSM with scalars
    v[j] = z'_{j mod 2} s[j], gamma = <v, L_tilde>, and Q_R is the (2R+1)-term MSM of
    _unfold_commitment.  Same accept/reject decision as the round-by-round verifier."""
    order = transcript.order
    N = len(g_hat)
    rounds, deferred = [], []
    m = N
    while True:
        A = _pt(proof["A" + str(round_i)])
        B = _pt(proof["B" + str(round_i)])
        c = transcript.round_challenge(round_i, A, B, None, k, Q, None)
        rounds.append(c)
        deferred.append((A, B, c))
        m //= 2
        if m <= 2:
            break
        round_i += 1
    z_prime = proof["z_prime"]
    low_bits = (len(z_prime) - 1).bit_length()
    if N != (1 << (len(rounds) + low_bits)) or len(z_prime) != (1 << low_bits):
        return False
    ctx = g_hat.ctx
    zp = ScalarVector.from_ints([pivot._residue(v) for v in z_prime], ctx)
    v = ScalarVector.empty(N, ctx)
    ctx.fr_challenge_products(rounds, low_bits, zp.ptr, v.ptr)
    gamma = v.dot_dev(_coeffs_dev(L_tilde))        # stays on the device: it is only ever the exponent of k below
    from .device import get_aux_context
    aux = get_aux_context()
    # the short one first: a (2R+1)-term MSM is ~25 dependent launches ending in a 0.2-ms recombination chain - pure
    # latency that now runs on its own stream while the host enqueues the N-term commitment and the GPU computes it
    # (3.26 -> 2.96 ms per verify at N = 2^20 on one box, alternating runs: before, the N-term one was back first)
    q_pending = _unfold_commitment(Q, deferred, order, aux, wait=False)
    pending = pivot._commit_launch(v, gamma, g_hat, k, ctx)
    return bool(pending.result() == q_pending.result())


def _final_check_host(g_prime, k, L_tilde, z_prime, order):
    """z'_0 g'_0 + z'_1 g'_1 + L'(z') k (compressed_pivot.py:193-195) for a device-resident g' of two elements: three
    ladders on the host (vmpc_ed25519_lincomb_host, 0.15 ms) instead of a three-term MSM through the bucket pipeline
    and a device inner product (2.7 + 0.5 ms at the end of every reference-transcript verify).  None: not this shape."""
    try:
        if not (isinstance(g_prime, PointVector) and len(g_prime) == len(z_prime) <= 2 and L_tilde.constant == 0):
            return None
        zs = [pivot._residue(v) for v in z_prime]
        co = L_tilde.coeffs.to_ints() if isinstance(L_tilde.coeffs, ScalarVector) else \
            [pivot._residue(v) for v in L_tilde.coeffs]
        if len(co) != len(zs):
            return None
        e = sum(a * b for a, b in zip(co, zs)) % order
        aff = g_prime.affine_array()
        pts = [aff[i].tobytes() for i in range(len(zs))] + [k.to_affine_bytes()]
        return Ed25519Point.from_affine_bytes(_native.lincomb_host(pts, zs + [e]))
    except (TypeError, ValueError, AttributeError):
        return None


def protocol_4_verifier(g_hat, k, Q, L_tilde, gf, proof, round_i=0, transcript=None, _checked=False):
    """Non-interactive Protocol 4, verifier (compressed_pivot.py:148-202).  Returns False (never
    raises) for a proof whose points are not elements of the order-l group."""
    g_hat = pivot._points_on_device(g_hat)
    if not _checked and not _valid_group_elements(
            _proof_points(proof) + ([] if isinstance(Q, (_LazyQ, _LazyPoint)) else [Q]), g_hat.ctx):
        return False
    k = _pt(k)
    Q = Q if isinstance(Q, (_LazyQ, _LazyPoint)) else _pt(Q)
    if not isinstance(transcript, _Transcript):
        transcript = _Transcript(transcript or "reference", k.order)
    if transcript.mode == "compact" and len(g_hat) >= 4 and (len(g_hat) & (len(g_hat) - 1)) == 0:
        return _protocol_4_verifier_compact(g_hat, k, Q, L_tilde, gf, proof, round_i, transcript)
    if transcript.mode == "reference" and isinstance(Q, _LazyQ):
        Q = Q.point()
    deferred = []            # compact mode: (A_i, B_i, c_i), Q unfolded once at the end
    while True:
        half = len(g_hat) // 2
        g_l, g_r = g_hat[:half], g_hat[half:]
        A = _pt(proof["A" + str(round_i)])
        B = _pt(proof["B" + str(round_i)])
        c = transcript.round_challenge(round_i, A, B, g_hat, k, Q, L_tilde, who="protocol_4_verifier")
        g_prime = g_l.fold(g_r, c, stream_text=transcript.mode == "reference" and half > 2)
        if transcript.mode == "reference":
            Q = _fold_commitment(A, Q, B, c)
        else:
            deferred.append((A, B, c))
        L_tilde = _fold_form(L_tilde, c, half, gf)
        if transcript.mode == "reference" and isinstance(L_tilde.coeffs, ScalarVector) and len(g_prime) > 2:
            # the next pre-image ends with L~: its text on the side stream now, not inside the next hash call (where it
            # was 5 + 1.3 + ... ms of waiting per proof, round 6)
            L_tilde.coeffs.text_begin()
        if len(g_prime) <= 2:
            z_prime = proof["z_prime"]
            Q_check = _final_check_host(g_prime, k, L_tilde, z_prime, transcript.order)
            if Q_check is None:
                Q_check = pivot.vector_commitment(z_prime, int(L_tilde(z_prime)), g_prime, k)
            logger_cp.debug("Arrived in final step of protocol_4_verifier.")
            if deferred:
                Q = _unfold_commitment(Q, deferred, transcript.order)
            if isinstance(Q, _LazyPoint):
                Q = Q.resolve()
            return bool(Q_check == Q)
        g_hat = g_prime
        round_i += 1


# ---- Protocol 5 ----------------------------------------------------------------------------------------

def _p5_challenges(mode, order, generators, t, A, P, L, y, who="protocol_5_prover"):
    if mode == "reference":
        # compressed_pivot.py:117-130
        input_list = [t, A.normalize(), generators, P.normalize(), L, y]
        pivot.log_hash_input(logger_cp_hin, who, input_list)
        tag = "First hash of compressed pivot"
        c0, c1 = pivot.fiat_shamir_hash_variants(input_list, [[0, tag], [1, tag]], order)
        return c0, c1, None
    seed = _compact_seed(generators, P, L, y, t, A)
    c0 = _challenge(hashlib.sha256(seed + b"\x00").digest(), order)
    c1 = _challenge(hashlib.sha256(seed + b"\x01").digest(), order)
    return c0, c1, seed


def _p5_setup(generators, k, seed, mode, order):
    if mode == "reference":
        return _Transcript(mode, order)
    # Q = A * P^c0 * k^(c1 (c0 y + t)) is a function of values the seed already binds
    state = hashlib.sha256(b"vmpc-ac20/p4/v2" + seed + k.to_affine_bytes()).digest()
    return _Transcript(mode, order, state)


class _LazyQ:
    """Q_0 = A * P^c0 * k^e kept as (scalar, point) terms: the compact prover never needs the
    point, the compact verifier folds the terms into its final MSM."""

    def __init__(self, A, P, k, c0, e, order):
        self.terms = [(1, A), (c0 % order, P), (e % order, k)]

    def point(self):
        # on the host in C (vmpc_ed25519_lincomb_host, ~0.15 ms; as Python big-int ladders 8 ms of the first round's
        # hash call): only the group element matters - the transcript hashes Q normalised (compressed_pivot.py:52)
        raw = _native.lincomb_host([pt.to_affine_bytes() for _, pt in self.terms], [sc for sc, _ in self.terms])
        return Ed25519Point.from_affine_bytes(raw)


def _extend_form(L, c1):
    """L~ = (L.coeffs || 0) * c1 (compressed_pivot.py:141)."""
    if isinstance(L.coeffs, ScalarVector):
        return pivot.LinearForm(L.coeffs.axpy_concat(c1, None, 0))
    return pivot.LinearForm(L.coeffs + [0]) * c1


def protocol_5_prover(generators, P, L, y, x, gamma, gf, transcript=None, r=None, rho=None):
    """Compressed Sigma-protocol Pi_c, prover (compressed_pivot.py:89-145).

    Extensions over the reference signature (all optional): `transcript` selects the
    Fiat-Shamir mode; `r`, `rho` supply the masks instead of drawing them from `prng`
    (used to keep 2^20 draws off the Python interpreter)."""
    mode = transcript or TRANSCRIPT
    g, h, k = generators["g"], _pt(generators["h"]), _pt(generators["k"])
    P = _pt(P)
    proof = {}
    n = len(x)
    L, y = pivot.affine_to_linear(L, y, n)
    assert bin(n + 1).count("1") == 1, \
        "This implementation requires n+1 to be power of 2 (else, use padding with zeros)."
    order = gf.order
    device_mode = _on_device(x, L.coeffs) or isinstance(r, ScalarVector)

    if r is None:
        r = list(prng.randrange(order) for i in range(n))
    if rho is None:
        rho = prng.randrange(order)
    if device_mode:
        x = pivot._as_device(x)
        r = pivot._as_device(r)
        L = pivot.AffineForm(_coeffs_dev(L), L.constant)
    gv = pivot._points_on_device(g)
    if mode == "reference" and device_mode and isinstance(g, PointVector):
        # the O(N) part of the first pre-image does not depend on A: format and copy it while
        # the announcement's MSM runs
        gv.text_begin()
        L.coeffs.text_begin()

    logger_cp.debug("Calculate t, A.")
    if mode == "compact" and device_mode and isinstance(r, ScalarVector) and isinstance(gv, PointVector) \
            and len(r):
        # the announcement's MSM first, everything the host can do without A behind it: the form's digest
        # (leaves hashed on a side stream, 8192 of them rehashed here at N = 2^20) and t = L(r), whose inner
        # product queues behind the MSM
        _form_digest_begin(L)
        pending = pivot._commit_launch(r, rho, gv, h, gv.ctx)
        L._form_digest = _form_digest(L)
        t = L(r)
        A = pending.result()
    else:
        if mode == "compact" and device_mode:
            _form_digest_begin(L)
        t = L(r)
        A = pivot.vector_commitment(r, rho, gv, h)
    if device_mode and isinstance(t, int):
        t = gf(t)
    proof["t"] = t
    proof["A"] = A

    gens_for_hash = {"g": g if isinstance(g, PointVector) or mode == "compact" else list(g),
                     "h": generators["h"], "k": generators["k"]}
    c0, c1, seed = _p5_challenges(mode, order, gens_for_hash, t, A, P, L, y)
    logger_cp_hout.debug(f"After hash, hash=\n{c0}, {c1}")

    phi = gf(c0 * gamma + rho)
    if device_mode:
        z_hat = x.axpy_concat(c0, r, phi)          # z and its extension in one pass
        z = z_hat[:n]
    else:
        z = [c0 * x_i + r[i] for i, x_i in enumerate(x)]
        z_hat = z + [phi]
    g_hat = gv + [h]
    logger_cp.debug("Calculate Q.")
    Q = _LazyQ(A, P, k, c0, int(c1 * (c0 * y + t)), order)
    if mode == "reference":
        Q = Q.point()
    L_tilde = _extend_form(L, c1)
    if mode == "reference" and isinstance(L_tilde.coeffs, ScalarVector):
        # round 0's pre-image ends with L~ (82 MB of text at N = 2^20): formatted and copied on the side stream while
        # the pair is computed and the 250 MB of generator text in front of it are hashed - left to the hash call it was
        # 8 ms of waiting inside it (round 6, scripts/ref_stall_probe.py)
        L_tilde.coeffs.text_begin()
    if not (mode == "compact" and device_mode):
        # compressed_pivot.py:142's self-check; on the compact device path it would be two more inner products
        # with a host round trip each (0.15 ms) for an identity that tests/test_gpu_protocol.py pins
        assert _same_residue(L(z) * c1, L_tilde(z_hat), order)

    return protocol_4_prover(g_hat, k, Q, L_tilde, z_hat, gf, proof,
                             transcript=_p5_setup(generators, k, seed, mode, order))


def protocol_5_verifier(generators, P, L, y, proof, gf, transcript=None):
    """Compressed Sigma-protocol Pi_c, verifier (compressed_pivot.py:205-239)."""
    mode = transcript or TRANSCRIPT
    g, h, k = generators["g"], _pt(generators["h"]), _pt(generators["k"])
    # on-curve / canonical now; the order-l ladder runs beside the rest and is joined before accepting (a proof
    # with a point outside the group gives garbage below, never an accept: the verdict is ANDed in)
    membership = _valid_group_elements_begin([P] + _proof_points(proof))
    if membership.verdict is False:
        return False
    P = _pt(P)
    order = gf.order
    n = len(g)
    L, y = pivot.affine_to_linear(L, y, n)
    if mode == "compact" and isinstance(L.coeffs, ScalarVector):
        _form_digest_begin(L)
    t = proof["t"]
    A = _pt(proof["A"])
    gens_for_hash = {"g": g if isinstance(g, PointVector) or mode == "compact" else list(g),
                     "h": generators["h"], "k": generators["k"]}
    c0, c1, seed = _p5_challenges(mode, order, gens_for_hash, t, A, P, L, y, who="protocol_5_verifier")
    logger_cp_hout.debug(f"After hash, hash=\n{c0}, {c1}")
    g_hat = pivot._points_on_device(g) + [h]
    Q = _LazyQ(A, P, k, c0, int(c1 * (c0 * y + t)), order)
    if mode == "reference":
        Q = Q.point()
    L_tilde = _extend_form(L, c1)
    if mode == "reference" and isinstance(L_tilde.coeffs, ScalarVector):
        L_tilde.coeffs.text_begin()              # (as in the prover: round 0's pre-image ends with L~)
    try:
        verdict = protocol_4_verifier(g_hat, k, Q, L_tilde, gf, proof,
                                      transcript=_p5_setup(generators, k, seed, mode, order), _checked=True)
    finally:
        in_group = membership.result()
    return bool(verdict) and in_group
