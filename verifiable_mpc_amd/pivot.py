"""Drop-in counterpart of verifiable_mpc/ac20/pivot.py (AC20 Protocol 2 and the
Pedersen vector commitment) with the group work on MI355X.

Same names, positional arguments, return shapes and error behaviour as the reference:
    list_mul              pivot.py:26-28
    AffineForm/LinearForm pivot.py:31-116
    _int                  pivot.py:119-128
    fiat_shamir_hash      pivot.py:131-136
    vector_commitment     pivot.py:139-145
    affine_to_linear      pivot.py:148-153
    prove_linear_form_eval / verify_linear_form_proof   pivot.py:156-205

Inputs may be the reference's Python lists (scalars stay Python objects and keep the
reference's typing semantics, points are sent to the GPU per call) or the device-resident
vectors of verifiable_mpc_amd.device (everything stays in HBM).
"""
import hashlib
import time
import logging
from random import SystemRandom

from . import _native
from .device import DeviceScalar, PointVector, ScalarVector, get_context, reduce_scalar
from .fields import FiniteFieldElement
from .groups import EllipticCurvePoint as EllipticCurveElement
from .groups import Ed25519Point, as_point

prng = SystemRandom()

logger_piv = logging.getLogger("pivot")
logger_piv.setLevel(logging.INFO)


class SecureObject:
    """Placeholder for mpyc.sectypes.SecureObject (pivot.py:17): secret-shared values are
    the MPyC driver's business and never reach this module."""


def list_mul(x):
    """Product of a list of group elements in the reference's tree order (pivot.py:26-28)."""
    if isinstance(x, PointVector):
        if not x.has_proj:
            raise ValueError("list_mul needs projective representatives")
        ctx = x.ctx
        out = ctx.alloc(96)
        ctx.tree_reduce(x.proj_ptr, len(x), True, out.ptr)
        return Ed25519Point.from_proj_bytes(ctx.download(out.ptr, 96).tobytes())
    return list_mul(PointVector.from_points(list(x)))


def _is_number(v):
    return isinstance(v, (int, FiniteFieldElement, SecureObject)) or _duck_field(v)


def _duck_field(v):
    # foreign field elements (e.g. real MPyC GF elements) quack like ours
    return hasattr(v, "value") and hasattr(type(v), "modulus") and not isinstance(v, EllipticCurveElement)


class AffineForm:
    """Dense affine form sum_i coeffs[i] * x_i + constant over the scalar field
    (pivot.py:31-95).  `coeffs` is either a Python list - element types are then kept exactly
    as the caller supplied them, like the reference does - or a device ScalarVector."""

    def __init__(self, coeffs, constant):
        self.coeffs, self.constant = coeffs, constant

    # which class a sum / shifted form has: the reference keeps type(self) for AffineForm and
    # degrades LinearForm sums to AffineForm (pivot.py:49, :116)
    def _sum_type(self):
        return type(self)

    def _device(self):
        return isinstance(self.coeffs, ScalarVector)

    def __len__(self):
        return len(self.coeffs)

    def __repr__(self):
        return f"{str(self.coeffs)}, {str(self.constant)}"

    def __eq__(self, other):
        # coefficients only, as pivot.py:78-79
        mine, theirs = self.coeffs, other.coeffs
        mine = mine.to_ints() if isinstance(mine, ScalarVector) else mine
        theirs = theirs.to_ints() if isinstance(theirs, ScalarVector) else theirs
        return mine == theirs

    def __add__(self, other):
        if isinstance(other, AffineForm):
            assert len(self) == len(other), "Length of linear forms to add not consistent."
            if self._device() or other._device():
                summed = _as_device(self.coeffs).axpy(1, _as_device(other.coeffs))
            else:
                summed = [a + b for a, b in zip(self.coeffs, other.coeffs)]
            return self._sum_type()(summed, self.constant + other.constant)
        if _is_number(other):
            return self._sum_type()(self.coeffs, self.constant + other)
        raise NotImplementedError(f"Addition of form not defined for type: {type(other)}")

    def __radd__(self, other):
        # lets sum([...forms...]) start from the int 0
        return self if (isinstance(other, int) and other == 0) else self.__add__(other)

    def __sub__(self, other):
        return self + (-1) * other

    def __mul__(self, factor):
        if not (isinstance(factor, (int, FiniteFieldElement)) or _duck_field(factor)):
            raise NotImplementedError(f"Multiplication of form not defined for type: {type(factor)}")
        if self._device():
            scaled = self.coeffs.scale(factor)
        else:
            scaled = [c * factor for c in self.coeffs]
        return type(self)(scaled, self.constant * factor)

    __rmul__ = __mul__

    def eval(self, values):
        assert len(values) == len(self.coeffs), \
            "Length of inputs to be equal to coefficients of linear form."
        if self._device() or isinstance(values, ScalarVector):
            inner = _as_device(self.coeffs).dot(_as_device(values))     # csrc/frvec.hip
            field = _field_of(self.constant)
            return (field(inner) if field else inner) + self.constant
        return sum([c * v for c, v in zip(self.coeffs, values)]) + self.constant

    __call__ = eval


class LinearForm(AffineForm):
    """Form without constant term (pivot.py:98-116): the constant argument is ignored and
    sums of linear forms come back as AffineForm."""

    def __init__(self, coeffs, constant=0):
        super().__init__(coeffs, 0)

    def _sum_type(self):
        return AffineForm


def _as_device(v):
    return v if isinstance(v, ScalarVector) else ScalarVector.from_ints([_residue(c) for c in v])


def _residue(v):
    return reduce_scalar(v.value if _duck_field(v) and not isinstance(v, FiniteFieldElement) else v)


def _field_of(v):
    return type(v) if isinstance(v, FiniteFieldElement) or _duck_field(v) else None


def _int(value):
    """pivot.py:119-128."""
    if isinstance(value, (int, SecureObject)):
        return value
    elif isinstance(value, FiniteFieldElement) or _duck_field(value):
        return int(value)
    else:
        raise NotImplementedError


# ---- Fiat-Shamir ---------------------------------------------------------------------------------

# What the reference-transcript modes spend in SHA-256 itself: bytes hashed and seconds inside update() since the last
# reset (bench.py prices prove / verify against this floor; pivot.py:131-136 hashes ~1 GB of text at N = 2^20).
HASH_STATS = {"bytes": 0, "seconds": 0.0, "calls": 0}


def hash_stats(reset=False):
    out = dict(HASH_STATS)
    if reset:
        HASH_STATS.update(bytes=0, seconds=0.0, calls=0)
    return out


class _CountingSha256:
    """hashlib.sha256 with a byte and time count on update() (large buffers only: the clock is read per call)"""
    __slots__ = ("h",)

    def __init__(self, h=None):
        self.h = h if h is not None else hashlib.sha256()

    def update(self, b):
        n = len(b)
        if n < 4096:
            self.h.update(b)
            HASH_STATS["bytes"] += n
            return
        t0 = time.perf_counter()
        self.h.update(b)
        HASH_STATS["seconds"] += time.perf_counter() - t0
        HASH_STATS["bytes"] += n
        HASH_STATS["calls"] += 1

    def copy(self):
        return _CountingSha256(self.h.copy())

    def digest(self):
        return self.h.digest()


def _feed(h, obj):
    """Stream str(obj) into the hash without materialising it: byte-identical to
    str(input_list).encode('utf-8') for the object kinds on the AC20 path; device vectors
    contribute text formatted by csrc/format.hip."""
    if isinstance(obj, (PointVector, ScalarVector)):
        h.update(b"[")
        if len(obj):
            for piece in obj.text_chunks():      # hashed piece by piece while the rest is still on the link
                h.update(memoryview(piece))
        h.update(b"]")
    elif isinstance(obj, AffineForm):
        _feed(h, obj.coeffs)
        h.update(b", ")
        _feed(h, obj.constant)
    elif type(obj) is list:
        h.update(b"[")
        for i, item in enumerate(obj):
            if i:
                h.update(b", ")
            _feed(h, item)
        h.update(b"]")
    elif type(obj) is dict:
        h.update(b"{")
        for i, (k, v) in enumerate(obj.items()):
            if i:
                h.update(b", ")
            h.update(repr(k).encode("utf-8") + b": ")
            _feed(h, v)
        h.update(b"}")
    else:
        h.update(repr(obj).encode("utf-8"))


class _LogSink:
    """update() target for _feed that writes the pre-image to a logger: one record per LOG_PIECE bytes, so that the
    ~250 MB text of a 2^20-generator round never exists as one Python string"""
    LOG_PIECE = 1 << 20

    def __init__(self, logger, method):
        self.logger, self.method, self.buf, self.part = logger, method, bytearray(), 0

    def update(self, b):
        self.buf += b
        while len(self.buf) >= self.LOG_PIECE:
            self._emit(self.buf[:self.LOG_PIECE])
            del self.buf[:self.LOG_PIECE]

    def _emit(self, piece):
        what = "Before fiat_shamir_hash, input_list=" if self.part == 0 else f"input_list continued (part {self.part})="
        self.logger.debug(f"Method {self.method}: {what}\n{bytes(piece).decode('utf-8')}")
        self.part += 1

    def close(self):
        if self.buf or self.part == 0:
            self._emit(self.buf)
        self.buf = bytearray()


def log_hash_input(logger, method, input_list):
    """The reference's hash-input dump (compressed_pivot.py:56-58,122-124,171-173,226-228: logger
    "compressed_pivot_hash_inputs" at DEBUG): `Method <name>: Before fiat_shamir_hash, input_list=\n<str(input_list)>`.
    Costs a second formatting of the pre-image, so nothing happens unless the logger is enabled for DEBUG."""
    if not logger.isEnabledFor(logging.DEBUG):
        return
    sink = _LogSink(logger, method)
    _feed(sink, input_list)
    sink.close()


def fiat_shamir_hash(input_list, order):
    """pivot.py:131-136: int.from_bytes(sha256(str(input_list)), 'little') % order."""
    h = _CountingSha256()
    _feed(h, input_list)
    return int.from_bytes(h.digest(), "little") % order


def fiat_shamir_hash_variants(common_items, tails, order):
    """[fiat_shamir_hash(common_items + tail, order) for tail in tails] with the shared
    prefix hashed once (SHA-256 state copied).  Protocol 5 hashes the same O(N) list twice,
    differing only in the trailing [0|1, tag] (compressed_pivot.py:125-130)."""
    h = _CountingSha256()
    h.update(b"[")
    for i, item in enumerate(common_items):
        if i:
            h.update(b", ")
        _feed(h, item)
    out = []
    for tail in tails:
        ht = h.copy()
        for j, item in enumerate(tail):
            if common_items or j:
                ht.update(b", ")
            _feed(ht, item)
        ht.update(b"]")
        out.append(int.from_bytes(ht.digest(), "little") % order)
    return out


# ---- Pedersen vector commitment ----------------------------------------------------------------

def _points_on_device(g):
    return g if isinstance(g, PointVector) else PointVector.from_points(list(g))


def _scalars_on_device(x):
    if isinstance(x, ScalarVector):
        return x
    return ScalarVector.from_ints([_residue(_int(x_i)) for x_i in x])


def _exact_commitment(x, gamma, gv, h):
    """The reference's own operation sequence (pivot.py:143-145) on the device:
    `g[i] ** _int(x_i)` per term (right-to-left ladder, a negative exponent inverts the base),
    `list_mul` (mpctools.reduce tree, identity appended), then `(h ** gamma) * prod` - so the
    un-normalised (X:Y:Z) of the result is the one the reference holds [mpyc-recall formulas]."""
    import numpy as np
    ctx = gv.ctx
    n = len(x)
    if not gv.has_proj:
        raise ValueError("exact_representative needs projective generators")
    terms = ctx.alloc(max(1, 96 * n))
    if isinstance(x, ScalarVector):
        ctx.repeat(gv.proj_ptr, n, False, x.ptr, n, 0, terms.ptr, None)       # residues as they are
    elif n:
        exps = [_int(v) for v in x]
        # plain Python ints are not residues: they keep their sign and their size (z' of a later
        # round is hundreds of bits longer, compressed_pivot.py:76); the ladder kernel takes
        # sign-magnitude exponents below 2^255, the few longer ones run on the host
        big = [i for i, e in enumerate(exps) if abs(e) >> 255]
        enc = b"".join(((abs(e) if not abs(e) >> 255 else 0) | ((1 << 255) if e < 0 and not abs(e) >> 255 else 0))
                       .to_bytes(32, "little") for e in exps)
        sc = ctx.upload(np.frombuffer(enc, np.uint8))
        ctx.repeat(gv.proj_ptr, n, False, sc.ptr, n, 2, terms.ptr, None)
        for i in big:
            term = Ed25519Point.repeat(gv[i], exps[i])
            ctx.upload_into(terms.ptr + 96 * i, np.frombuffer(term.to_proj_bytes(), np.uint8))
    out = ctx.alloc(96)
    ctx.tree_reduce(terms.ptr, n, True, out.ptr)
    prod = Ed25519Point.from_proj_bytes(ctx.download(out.ptr, 96).tobytes())
    return Ed25519Point.operation(Ed25519Point.repeat(h, _int(gamma)), prod)


def _wants_exact(x, gv):
    """List inputs are the reference's calling convention (demo_zkp_ac20.py, circuit_sat_cb.py): there
    the caller may hash or print the commitment WITHOUT normalising it (circuit_sat_cb.py:107-111,
    demo_zkp_ac20.py:84), so the representative is part of the contract.  Device vectors are the
    N = 2^20 convention: group element only, Pippenger."""
    return not isinstance(x, ScalarVector) and gv.has_proj


def vector_commitment(x, gamma, g, h, exact_representative=None):
    """Pedersen vector commitment, Definition 1 of AC20 (pivot.py:139-145):
    h^gamma * prod_i g_i^{x_i}.

    `x` a Python list (the reference's convention): the reference's per-term `**` and reduce tree
    are replayed on the device, and the returned element has the same un-normalised (X:Y:Z) the
    reference would hold - bit-identical transcripts for callers that hash it as it is
    (circuit_sat_cb.py:107).  `x` a device ScalarVector: one (len(x)+1)-term Pippenger MSM, the
    result affine-normalised (Z = 1).  `exact_representative` forces either path."""
    assert len(g) >= len(x), "Not enough generators."
    n = len(x)
    gv = _points_on_device(g)
    if exact_representative is None:
        exact_representative = _wants_exact(x, gv)
    if exact_representative:
        return _exact_commitment(x, gamma, gv[:n] if len(gv) != n else gv, _as_point(h))
    xs = _scalars_on_device(x)
    return _commit_launch(xs, gamma, gv, _as_point(h), gv.ctx).result()


_as_point = as_point       # ours, or a foreign three-coordinate element converted (groups.as_point)


class _PendingCommitment:
    """An MSM enqueued on a context; result() synchronises that context and fetches the point.  `relaunch`
    enqueues the same commitment again: used once, on the general path, when the fused short path reports scalars
    beyond its capacities (VMPC_E_AGAIN, csrc/msm_short.hip)."""

    def __init__(self, ctx, out, keepalive, relaunch=None):
        self.ctx, self.out, self.keepalive, self.relaunch = ctx, out, keepalive, relaunch

    def result(self):
        try:
            self.ctx.sync()
            raw = self.ctx.download(self.out.ptr, 96).tobytes()
            again = _void_point(raw)        # this commitment overflowed, another caller collected the status word
        except _native.VmpcError as e:
            if e.code != _native.E_AGAIN:
                raise
            # the status word is one per context: the overflow may be another pending commitment's.  The sync has
            # completed; this one's own marker decides
            raw = self.ctx.download(self.out.ptr, 96).tobytes()
            again = _void_point(raw)
        if again:
            if self.relaunch is None:
                raise _native.VmpcError(_native.E_AGAIN, "vector_commitment")
            self.ctx.on_general_path(lambda: (self.relaunch(), self.ctx.sync()))
            raw = self.ctx.download(self.out.ptr, 96).tobytes()
        # the kernel leaves the sum in extended coordinates; the one field inversion of
        # .normalize() is O(1) host glue (25 us of big-int pow vs a 120 us single-lane chain)
        self.keepalive = None
        return Ed25519Point.from_proj_bytes(raw).normalize()


def _void_point(raw):
    """Z = 0: what the fused short path writes for a commitment whose scalars were beyond its capacities
    (csrc/msm_short.hip, csrc/msm_reduce_tree.hip `poison`) - no point of the curve has it"""
    return raw[64:96] == bytes(32)


def _table_args(xs, gamma, gv, h, ctx):
    """(table, number of main scalars, extras-scalar buffer) when the commitment can run on gv's fixed-base
    table: gv is (a prefix of) a tabulated vector, possibly followed by the first `tail` extras of the table
    (g + [h]), and h is one of the remaining extras.  None otherwise."""
    import numpy as np
    # (a vector that holds the wide-window table beside its 16-bit-window one - PointVector.precompute(wide=True) -
    # is committed to over the former)
    table = getattr(gv, "_wide", None) or getattr(gv, "_table", None)
    if table is None:
        return None
    n = len(xs)
    tail = gv._table_tail
    n_main = len(gv) - tail
    slot = table.extra_index(h)
    used_tail = max(0, n - n_main)
    if slot is None or slot < used_tail:
        return None
    esc = bytearray(32 * len(table.extra_bytes))
    if not isinstance(gamma, DeviceScalar):
        esc[32 * slot:32 * slot + 32] = reduce_scalar(_int(gamma)).to_bytes(32, "little")
    gam = ctx.upload(np.frombuffer(bytes(esc), np.uint8))
    if isinstance(gamma, DeviceScalar):
        ctx.copy(gam.ptr + 32 * slot, gamma.ptr, 32)
    if used_tail:
        ctx.copy(gam.ptr, xs.ptr + 32 * n_main, 32 * used_tail)
    return table, min(n, n_main), gam


# A generator vector somebody commits to a SECOND time is a CRS (pivot.py:139-145 is called with the same g, h for every
# proof): it is tabulated then, without being asked (PointVector.precompute), and every later commitment over it skips
# the point preparation and most or all of the window recombination.  The form with the shortest latency for one
# commitment alone (scripts/rows20_probe.py, round 6): 16 rows below 2^19 generators (<= 1 GiB; up to 2^17 columns the
# fused short path), the 13-row wide-window table from 2^19 up (0.66 / 0.98 / 1.69 ms at 2^19 / 2^20 / 2^21 against
# 0.71 / 1.04 / 2.06 for the best 16-bit-window table) while it fits AUTO_TABLE_BUDGET; at most AUTO_TABLE_TOTAL bytes
# of such tables alive per process (288 GB of HBM: these caps are politeness, not necessity).
AUTO_TABLE_MIN = 1 << 10
AUTO_TABLE_BUDGET = 8 << 30
AUTO_TABLE_TOTAL = 32 << 30
AUTO_TABLE_WIDE_MIN = 1 << 19
_auto_table_bytes = [0]


def _auto_table_rows(n_points):
    """(rows, bytes) of the table _auto_tabulate builds over n_points generators + one extra"""
    cols = n_points + 1
    if n_points >= AUTO_TABLE_WIDE_MIN:
        stride = (cols + 8191) & ~8191
        if stride <= 1 << 22 and 13 * 128 * stride <= AUTO_TABLE_BUDGET:
            return 13, 13 * 128 * stride
    rows = 16
    while rows > 1 and rows * 128 * cols > 1 << 30:
        rows //= 2
    return rows, rows * 128 * cols


def _auto_tabulate(gv, h):
    if gv._table is not None or not isinstance(gv, PointVector) or len(gv) < AUTO_TABLE_MIN:
        return
    gv._commit_uses = getattr(gv, "_commit_uses", 0) + 1
    if gv._commit_uses != 2:
        return
    rows, nbytes = _auto_table_rows(len(gv))
    if _auto_table_bytes[0] + nbytes > AUTO_TABLE_TOTAL:
        return
    import weakref
    try:
        gv.precompute([h], rows=rows)
    except _native.VmpcError as e:
        # precompute synchronises the context: a short-path overflow of an EARLIER, not yet collected commitment
        # surfaces here (one status word per context).  That commitment finds its own void marker in result(); the
        # table build itself never takes the short path and has completed with the sync
        if e.code != _native.E_AGAIN:
            raise
        gv.ctx.sync()
        if gv._table is None:
            gv.precompute([h], rows=rows)
    _auto_table_bytes[0] += nbytes
    weakref.finalize(gv._table, _auto_table_released, nbytes)


def _auto_table_released(nbytes):
    _auto_table_bytes[0] -= nbytes


def _commit_launch(xs, gamma, gv, h, ctx):
    """enqueue h^gamma * prod g_i^{x_i} on `ctx` (device vectors in, 64-byte affine out)"""
    import numpy as np
    n = len(xs)
    out = ctx.alloc(128)
    if ctx is gv.ctx:
        _auto_tabulate(gv, h)
    targs = _table_args(xs, gamma, gv, h, ctx)
    if targs is not None:
        table, m, gam = targs

        def launch():
            ctx.msm_table(table.ptr, table.n, len(table.extra_bytes), xs.ptr, m, gam.ptr, out.ptr, None, rows=table.rows)
        launch()
        return _PendingCommitment(ctx, out, (gam, xs, gv, table), launch)
    if isinstance(gamma, DeviceScalar):
        gam = gamma
    else:
        gam = ctx.upload(np.frombuffer(reduce_scalar(_int(gamma)).to_bytes(32, "little"), np.uint8))
    hb = ctx.upload(np.frombuffer(h.to_affine_bytes(), np.uint8))
    ctx.msm(xs.ptr, gv.affine_ptr, n, gam.ptr, hb.ptr, 1, out.ptr, None)
    return _PendingCommitment(ctx, out, (gam, hb, xs, gv))


def vector_commitment_pair(x_a, gamma_a, g_a, x_b, gamma_b, g_b, h):
    """Two independent commitments with the same h (A_i and B_i of a Protocol-4 round,
    compressed_pivot.py:41-42) on two streams of the same GPU."""
    from .device import get_aux_context
    assert len(g_a) >= len(x_a) and len(g_b) >= len(x_b), "Not enough generators."
    gva, gvb = _points_on_device(g_a), _points_on_device(g_b)
    if _wants_exact(x_a, gva) and _wants_exact(x_b, gvb):
        # list mode: the proof's A_i, B_i carry the reference's representatives
        return vector_commitment(x_a, gamma_a, gva, h), vector_commitment(x_b, gamma_b, gvb, h)
    xa, xb = _scalars_on_device(x_a), _scalars_on_device(x_b)
    if getattr(gva, "_table", None) is not None and getattr(gvb, "_table", None) is gva._table:
        # both over the same tabulated CRS: ONE pass for the pair (vmpc_msm_table_batch_dev) - the bucket
        # reduction and the window recombination, latency chains, run once instead of twice side by side
        ctx = gva.ctx
        ta, tb = _table_args(xa, gamma_a, gva, h, ctx), _table_args(xb, gamma_b, gvb, h, ctx)
        if ta is not None and tb is not None and ta[1] == tb[1]:
            table, m = ta[0], ta[1]
            out = ctx.alloc(256)

            def launch():
                ctx.msm_table_batch(table.ptr, table.n, len(table.extra_bytes), [xa.ptr, xb.ptr], m,
                                    [ta[2].ptr, tb[2].ptr], out.ptr, None, rows=table.rows)
                ctx.sync()
            try:
                launch()
                raw = ctx.download(out.ptr, 256).tobytes()
                again = _void_point(raw[:96]) or _void_point(raw[128:224])
            except _native.VmpcError as e:
                if e.code != _native.E_AGAIN:
                    raise
                again = True
            if again:
                ctx.on_general_path(launch)
                raw = ctx.download(out.ptr, 256).tobytes()
            return (Ed25519Point.from_proj_bytes(raw[:96]).normalize(),
                    Ed25519Point.from_proj_bytes(raw[128:224]).normalize())
    main, aux = gva.ctx, get_aux_context()
    aux.wait_for(main)                 # inputs were produced on the main stream
    try:
        pa = _commit_launch(xa, gamma_a, gva, h, main)
        pb = _commit_launch(xb, gamma_b, gvb, h, aux)
        return pa.result(), pb.result()
    except BaseException:
        # a failed result() (e.g. NONCANON) must not hand this call's buffers back to the block cache
        # while the other stream still reads them
        for c in (main, aux):
            try:
                c.sync()
            except Exception:
                pass
        raise


def affine_to_linear(L, y, n):
    """pivot.py:148-153."""
    if isinstance(L.coeffs, ScalarVector):
        constant = L.constant          # L(zeros) == constant
    else:
        zeros = [0] * n
        constant = L(zeros)
    L_linear = L - constant
    y_linear = y - constant
    return L_linear, y_linear


def _pis_challenge(t, A, g, h, P, L, y, order):
    """Challenge of Pi_s: hash of [t, A, g, h, P, L, y] with A and P normalised when they
    are curve points (pivot.py:169-174, :194-201)."""
    if isinstance(A, EllipticCurveElement):
        A, P = A.normalize(), P.normalize()
    return fiat_shamir_hash([t, A, g, h, P, L, y], order)


def prove_linear_form_eval(g, h, P, L, y, x, gamma, gf):
    """Sigma protocol Pi_s (Protocol 2 of AC20), non-interactive (pivot.py:156-181):
    returns (z, phi, c)."""
    n = len(x)
    L, y = affine_to_linear(L, y, n)
    order = gf.order
    r = [gf(prng.randrange(order)) for _ in range(n)]      # masks: r first, then rho
    rho = prng.randrange(order)
    t = L(r)
    A = vector_commitment(r, rho, g, h)
    logger_piv.debug(f"Prover computed A={A}.")
    c = _pis_challenge(t, A, g, h, P, L, y, order)
    z = [c * x_i + r_i for x_i, r_i in zip(x, r)]
    phi = (c * gamma + rho) % order
    return z, phi, c


def verify_linear_form_proof(g, h, P, L, y, z, phi, c):
    """Verifier of Pi_s (pivot.py:184-205): recompute the announcement from the response
    and compare challenges."""
    L, y = affine_to_linear(L, y, len(z))
    P = as_point(P)
    # A = commit(z, phi) / P^c
    P_c_inv = Ed25519Point.inversion(Ed25519Point.repeat(P, int(c)))
    A_check = Ed25519Point.operation(vector_commitment(z, phi, g, h), P_c_inv)
    t_check = L(z) - c * y
    logger_piv.debug(f"Verifier computed A_check={A_check}, t_check={t_check}.")
    return bool(c == _pis_challenge(t_check, A_check, g, h, P, L, y, type(t_check).order))
