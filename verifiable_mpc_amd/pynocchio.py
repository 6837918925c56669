"""Pinocchio prover's group work on MI355X (BASELINE config 5).

    compute_proof    verifiable_mpc/trinocchio/pynocchio.py:228-273

The reference builds eight proof elements, each
    apply_to_list(point_add, [int(c[i]) * evalkey[<name i>] for i in qap.indices_mid])
(seven over G1, one over the twist, plus h(s) over the powers of s) and, in the
zero-knowledge case, adds `delta * evalkey[<t term>]`.  Here every element is ONE BN-256 MSM
(csrc/bn256.hip) with the zero-knowledge terms appended as extra (scalar, point) pairs.
Keys and proofs hold `BN256Point` / `BN256TwistPoint` objects (affine coordinates); foreign
points (e.g. MPyC's Jacobian elements) are accepted if they expose `.normalize()` and three
indexable coordinates.  Key generation, QAP construction and the pairing-based verifier stay
with the reference (out of scope).
"""
import numpy as np

from . import _native
from .device import get_context

P = 65000549695646603732796438742359905742825358107623003571877145026864184071783
ORDER = 65000549695646603732796438742359905742570406053903786389881062969044166799969


class BN256Point:
    """Affine point of G1 (y^2 = x^3 + 3 over F_p); `coords is None` is the point at infinity."""
    group = 1
    width = 64

    def __init__(self, coords=None):
        self.coords = None if coords is None else tuple(int(v) % P for v in self._flat(coords))

    @staticmethod
    def _flat(coords):
        return coords

    def to_bytes(self):
        if self.coords is None:
            return bytes(self.width)
        return b"".join(v.to_bytes(32, "little") for v in self.coords)

    @classmethod
    def from_bytes(cls, b):
        vals = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(cls.width // 32)]
        obj = cls.__new__(cls)
        obj.coords = None if not any(vals) else tuple(vals)
        return obj

    def normalize(self):
        return self

    def __eq__(self, other):
        return type(other) is type(self) and self.coords == other.coords

    def __hash__(self):
        return hash((self.group, self.coords))

    def __repr__(self):
        return "O" if self.coords is None else repr(list(self.coords))


class BN256TwistPoint(BN256Point):
    """Affine point of the twist over F_p[i]/(i^2+1): coords = (x.re, x.im, y.re, y.im)."""
    group = 2
    width = 128

    @staticmethod
    def _flat(coords):
        if len(coords) == 2:            # ((x.re, x.im), (y.re, y.im))
            return (coords[0][0], coords[0][1], coords[1][0], coords[1][1])
        return coords


def _as_bytes(pt):
    if isinstance(pt, BN256Point):
        return pt.group, pt.to_bytes()
    # foreign Jacobian / affine element: normalise and read x, y (each an int or a pair)
    q = pt.normalize() if hasattr(pt, "normalize") else pt
    x, y = q[0], q[1]
    if hasattr(x, "__len__") or (hasattr(x, "value") and hasattr(x.value, "__len__")):
        xs = list(x.value) if hasattr(x, "value") else list(x)
        ys = list(y.value) if hasattr(y, "value") else list(y)
        vals = [int(xs[0]), int(xs[1]), int(ys[0]), int(ys[1])]
        return 2, b"".join((v % P).to_bytes(32, "little") for v in vals)
    return 1, (int(x) % P).to_bytes(32, "little") + (int(y) % P).to_bytes(32, "little")


def msm(scalars, points, ctx=None):
    """sum_i scalars[i] * points[i] on the GPU; all points from the same group (G1 or twist)."""
    assert len(scalars) == len(points)
    ctx = ctx or get_context()
    if not points:
        raise ValueError("empty sum has no group")
    enc = [_as_bytes(p) for p in points]
    group = enc[0][0]
    assert all(g == group for g, _ in enc), "mixed groups in one sum"
    width = 64 if group == 1 else 128
    pts = np.frombuffer(b"".join(b for _, b in enc), dtype=np.uint8).reshape(-1, width)
    sc = _native.ints_to_array([int(s) % ORDER for s in scalars], 32)
    dp, ds, out = ctx.upload(pts), ctx.upload(sc), ctx.alloc(width)
    if ctx.bn256_validate(group, dp.ptr, len(points)):
        raise _native.VmpcError(_native.E_NOTONCURVE, "pynocchio.msm")
    ctx.bn256_msm(group, ds.ptr, dp.ptr, len(points), out.ptr)
    ctx.sync()
    cls = BN256Point if group == 1 else BN256TwistPoint
    return cls.from_bytes(ctx.download(out.ptr, width).tobytes())


# the eight proof elements of pynocchio.py:228-273: name -> (evalkey name of term i, zero-knowledge terms
# as (delta attribute, evalkey name))
_ELEMENTS = {
    "r_v*v_mid*g1": (lambda i: "r_v*v" + str(i) + "*g1", (("v", "r_v*t*g1"),)),
    "r_w*w_mid*g2": (lambda i: "r_w*w" + str(i) + "*g2", (("w", "r_w*t*g2"),)),
    "r_y*y_mid*g1": (lambda i: "r_y*y" + str(i) + "*g1", (("y", "r_y*t*g1"),)),
    "r_v*alpha_v*v_mid*g1": (lambda i: f"r_v*alpha_v*v{i}*g1", (("v", "r_v*alpha_v*t*g1"),)),
    "r_w*alpha_w*w_mid*g1": (lambda i: f"r_w*alpha_w*w{i}*g1", (("w", "r_w*alpha_w*t*g1"),)),
    "r_y*alpha_y*y_mid*g1": (lambda i: f"r_y*alpha_y*y{i}*g1", (("y", "r_y*alpha_y*t*g1"),)),
    "r_v*beta*v_mid+r_w*beta*w_mid+r_y*beta*y_mid*g1": (
        lambda i: f"r_v*beta*v+r_w*beta*w+r_y*beta*y{i}_g1",
        (("v", "r_v*beta*t*g1"), ("w", "r_w*beta*t*g1"), ("y", "r_y*beta*t*g1"))),
}


def _from_jacobian(group, raw):
    """(X : Y : Z) canonical residues -> affine BN256Point / BN256TwistPoint (x = X/Z^2, y = Y/Z^3)"""
    v = [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(raw) // 32)]
    if group == 1:
        X, Y, Z = v
        if Z == 0:
            return BN256Point(None)
        zi = pow(Z, P - 2, P)
        return BN256Point((X * zi * zi % P, Y * zi * zi % P * zi % P))
    mul = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    X, Y, Z = (v[0], v[1]), (v[2], v[3]), (v[4], v[5])
    if Z == (0, 0):
        return BN256TwistPoint(None)
    d = pow(Z[0] * Z[0] + Z[1] * Z[1], P - 2, P)
    zi = (Z[0] * d % P, (-Z[1] * d) % P)
    zi2 = mul(zi, zi)
    x, y = mul(X, zi2), mul(mul(Y, zi2), zi)
    return BN256TwistPoint((x[0], x[1], y[0], y[1]))


def _from_jacobian_many_g1(raws):
    """several G1 sums (X : Y : Z) -> affine, ONE modular inversion for all of them (Montgomery's trick)"""
    vals = [[int.from_bytes(r[32 * i:32 * i + 32], "little") for i in range(3)] for r in raws]
    zs = [v[2] for v in vals if v[2]]
    prefix, run = [], 1
    for z in zs:
        prefix.append(run)
        run = run * z % P
    inv = pow(run, P - 2, P) if zs else 0
    zinv = [0] * len(zs)
    for i in range(len(zs) - 1, -1, -1):
        zinv[i] = inv * prefix[i] % P
        inv = inv * zs[i] % P
    out, j = [], 0
    for X, Y, Z in vals:
        if Z == 0:
            out.append(BN256Point(None))
            continue
        zi = zinv[j]
        j += 1
        zi2 = zi * zi % P
        out.append(BN256Point((X * zi2 % P, Y * zi2 % P * zi % P)))
    return out


class _KeyVector:
    """One evaluation-key vector on the device: points uploaded, validated and tabulated once."""

    def __init__(self, ctx, points):
        enc = [_as_bytes(p) for p in points]
        group = enc[0][0]
        assert all(g == group for g, _ in enc), "mixed groups in one key vector"
        width = 64 if group == 1 else 128
        pts = np.frombuffer(b"".join(b for _, b in enc), dtype=np.uint8).reshape(-1, width)
        self._init_device(ctx, group, ctx.upload(pts), len(points))

    @classmethod
    def from_device(cls, ctx, group, points_buf, n):
        """from n affine points already in device memory (64 / 128 bytes each)"""
        self = cls.__new__(cls)
        self._init_device(ctx, group, points_buf, n)
        return self

    def _init_device(self, ctx, group, dp, n):
        self.group, self.n = group, n
        self.width = 64 if group == 1 else 128
        if ctx.bn256_validate(group, dp.ptr, n):
            raise _native.VmpcError(_native.E_NOTONCURVE, "pynocchio.PreparedKey")
        self.table = ctx.bn256_table_build(group, dp.ptr, n)
        ctx.sync()

    def launch(self, ctx, head, head_n, tail=()):
        """enqueue sum over the first head_n + len(tail) points: `head` is a device buffer of head_n
        scalars shared by several sums, `tail` a few host ints that follow them.  -> pending handle"""
        m = head_n + len(tail)
        assert m <= self.n
        jw = 3 * self.width // 2
        ds, out = ctx.alloc(max(32, 32 * m)), ctx.alloc(jw)
        if head_n:
            ctx.copy(ds.ptr, head.ptr, 32 * head_n)
        if tail:
            ctx.upload_into(ds.ptr + 32 * head_n, _native.ints_to_array([int(s) % ORDER for s in tail], 32))
        ctx.bn256_table_msm(self.group, self.table.ptr, self.n, ds.ptr, m, None, out.ptr)
        return ctx, out, jw, ds

    def result(self, pending):
        ctx, out, jw, _ = pending
        ctx.sync()
        # the sum comes back in Jacobian coordinates; the one inversion is O(1) host glue
        return _from_jacobian(self.group, ctx.download(out.ptr, jw).tobytes())

    def msm(self, ctx, scalars):
        head = ctx.upload(_native.ints_to_array([int(s) % ORDER for s in scalars], 32)) if scalars else None
        return self.result(self.launch(ctx, head, len(scalars)))


class PreparedKey:
    """The evaluation key of one circuit, prepared for many proofs: the eight point vectors that
    compute_proof reads (pynocchio.py:228-246) are converted, uploaded, checked and expanded into
    fixed-base tables ONCE; each proof then only ships its scalars.  Pass it to compute_proof in place
    of the evalkey dict."""

    def __init__(self, qap, evalkey, ctx=None):
        self.ctx = ctx or get_context()
        self.mid = list(qap.indices_mid)
        self.mid_index = np.asarray(self.mid, dtype=np.int64)
        self.vectors = {}
        self.zk_missing = {}      # element -> names of zero-knowledge key points the evalkey lacks
        for name, (key_fmt, zk) in _ELEMENTS.items():
            points = [evalkey[key_fmt(i)] for i in self.mid]
            missing = [zname for _, zname in zk if zname not in evalkey]
            if missing:
                self.zk_missing[name] = missing
            points += _tail_points(name, zk, evalkey, bool(missing))
            self.vectors[name] = _KeyVector(self.ctx, points)
        powers = []
        while "s^" + str(len(powers)) + "*g1" in evalkey:
            powers.append(evalkey["s^" + str(len(powers)) + "*g1"])
        self.vectors["h*g1"] = _KeyVector(self.ctx, powers)

    @classmethod
    def synthetic(cls, ctx, n, seed=3):
        """A key of n `mid` wires whose every entry is a distinct multiple of its group's generator, built on the
        device: the shape of a real prepared key without a circuit (bench.py, scripts/pinocchio_probe.py - timing
        only; parity of compute_proof is pinned on the reference-made fixture)."""
        g1 = (1).to_bytes(32, "little") + (P - 2).to_bytes(32, "little")
        g2 = b"".join(v.to_bytes(32, "little") for v in (
            64746500191241794695844075326670126197795977525365406531717464316923369116492,
            21167961636542580255011770066570541300993051739349375019639421053990175267184,
            17778617556404439934652658462602675281523610326338642107814333856843981424549,
            20666913350058776956210519119118544732556678129809273996262322366050359951122))
        rng = np.random.default_rng(seed)
        key = cls.__new__(cls)
        key.ctx, key.mid, key.vectors = ctx, list(range(n)), {}
        key.mid_index, key.zk_missing = np.arange(n), {}
        for name in list(_ELEMENTS) + ["h*g1"]:
            grp, gen, width = (2, g2, 128) if name.endswith("g2") else (1, g1, 64)
            zk = _ELEMENTS[name][1] if name in _ELEMENTS else ()
            tail = _tail_slots(name, zk)
            ex = rng.integers(0, 256, size=(n + len(tail), 32), dtype=np.uint8)
            ex[:, 31] &= 0x7F
            dg, de = ctx.upload(np.frombuffer(gen, np.uint8)), ctx.upload(ex)
            pts = ctx.alloc(width * (n + len(tail)))
            ctx.bn256_fixed_base(grp, dg.ptr, de.ptr, n + len(tail), pts.ptr)
            for j, used in enumerate(tail):          # columns of deltas this element does not use: infinity
                if not used:
                    ctx.upload_into(pts.ptr + width * (n + j), np.zeros(width, np.uint8))
            ctx.sync()
            key.vectors[name] = _KeyVector.from_device(ctx, grp, pts, n + len(tail))
        return key


# The six G1 sums over c_mid (pynocchio.py:229-246) go through ONE multi-key pass (vmpc_bn256_table_msm_multi_dev):
# they share the scalar vector c_mid || (delta_v, delta_w, delta_y), so each of their key vectors carries three
# trailing columns in that order - the element's zero-knowledge point where it uses the delta, the point at infinity
# where it does not (an infinity entry adds nothing, csrc/bn256_curve.h).
_DELTAS = ("v", "w", "y")
_SHARED_G1 = tuple(name for name in _ELEMENTS if name.endswith("g1"))


def _tail_slots(name, zk):
    """for each trailing column of the element's key vector: does the element use that delta?"""
    if name in _SHARED_G1:
        used = {attr for attr, _ in zk}
        return [d in used for d in _DELTAS]
    return [True] * len(zk)


def _tail_points(name, zk, evalkey, missing):
    if name in _SHARED_G1:
        by_delta = {attr: zname for attr, zname in zk}
        return [evalkey[by_delta[d]] if (d in by_delta and not missing) else BN256Point(None) for d in _DELTAS]
    return [] if missing else [evalkey[zname] for _, zname in zk]


def scalars_to_array(values):
    """Python ints / field elements -> (n, 32) uint8 canonical residues mod the group order, as fast as the
    interpreter allows (one to_bytes per element; the reduction only for values that need it).  (n, 32) uint8
    arrays pass through untouched: a caller that keeps its witness in numpy pays nothing here - at 2^18 terms
    this conversion (2 x 17 ms for c and h) is otherwise longer than the eight sums on the GPU."""
    if isinstance(values, np.ndarray):
        return _native.as_bytes_array(values, 32)
    vals = values if isinstance(values, list) else list(values)
    if not vals:
        return np.zeros((0, 32), np.uint8)
    if type(vals[0]) is not int:
        vals = [int(v) for v in vals]
    try:
        raw = b"".join([v.to_bytes(32, "little") for v in vals])       # (negatives and values >= 2^256 raise)
    except AttributeError:                                             # field elements after a leading int
        vals = [int(v) for v in vals]
        return scalars_to_array(vals)
    except OverflowError:
        raw = b"".join([(v % ORDER).to_bytes(32, "little") for v in vals])
        return np.frombuffer(raw, np.uint8).reshape(-1, 32)
    arr = np.frombuffer(raw, np.uint8).reshape(-1, 32)
    big = np.nonzero(arr[:, 31] >= 0x8f)[0]          # order = 0x8fb5...: only these rows can be >= order
    if len(big):
        arr = arr.copy()
        for i in big:
            arr[i] = np.frombuffer((vals[i] % ORDER).to_bytes(32, "little"), np.uint8)
    return arr


def _compute_proof_prepared(key, c, h, deltas):
    """The eight sums over a prepared key.  The shared `c_mid` scalars are converted and uploaded once; the SIX G1 sums
    over them are one multi-key pass (one recoding, sort and plan; six bucket launches over the one sorted index
    list; one reduction and one finishing launch), the twist sum over them runs beside it on a second stream; ONLY
    THEN are h's coefficients converted - on the host, while the GPU works through those seven - and the last sum
    enqueued.  The sums come back in Jacobian coordinates; their inversions are shared on the host (one modular
    inversion for all seven G1 results).
    c: indexable by qap.indices_mid (the reference's list of ints / field elements), or an (n_wires, 32) uint8
    array of canonical residues (rows taken by index); h: the reference's polynomial (.coeffs), a list, or an
    (len, 32) uint8 array."""
    from .device import get_aux_context
    ctx = key.ctx
    n_mid = len(key.mid)
    if deltas is not None and key.zk_missing:
        # the dict path fails with KeyError on the first absent name (pynocchio.py:229-246)
        raise KeyError("zero-knowledge deltas given but the prepared key lacks "
                       + ", ".join(sorted(n for names in key.zk_missing.values() for n in names)))
    if isinstance(c, np.ndarray):
        c_all = _native.as_bytes_array(c, 32)
        # (a key whose mid wires are ALL the wires, in order, needs no gather: 8 MB less to copy at 2^18 terms)
        c_mid = c_all if _mid_is_identity(key, len(c_all)) else np.ascontiguousarray(c_all[key.mid_index])
    else:
        c_mid = scalars_to_array([c[i] for i in key.mid])
    dvals = [int(getattr(deltas, d)) % ORDER for d in _DELTAS] if deltas is not None else []
    # c_mid || (delta_v, delta_w, delta_y): the scalar vector of the six G1 sums (and, through its own tail, the twist's)
    n_shared = n_mid + len(dvals)
    head = ctx.alloc(max(32, 32 * n_shared))
    if n_mid:
        ctx.upload_into(head.ptr, c_mid)
    if dvals:
        ctx.upload_into(head.ptr + 32 * n_mid, _native.ints_to_array(dvals, 32))
    hv = key.vectors["h*g1"]
    h_ctx = get_aux_context(21)
    h_coeffs = h if isinstance(h, (np.ndarray, list)) else h.coeffs
    g1 = [key.vectors[name] for name in _SHARED_G1]
    out_g1 = ctx.alloc(96 * len(g1))
    # three streams: the twist sum (the longest single one) first, the six-sum pass beside it, h's sum on a third -
    # the bucket kernels take turns on the chip, the reductions and recombinations (latency chains) overlap them
    twist_ctx = get_aux_context(20)
    twist_ctx.wait_for(ctx)               # `head` was filled on the main stream
    pending_twist = None
    for name, (_, zk) in _ELEMENTS.items():
        if name not in _SHARED_G1:
            tail = [int(getattr(deltas, attr)) for attr, _ in zk] if deltas is not None else []
            pending_twist = (name, key.vectors[name].launch(twist_ctx, head, n_mid, tail))
    ctx.bn256_table_msm_multi(1, [v.table.ptr for v in g1], g1[0].n, head.ptr, n_shared, out_g1.ptr)
    # h's coefficients: converted (the reference's ints) and uploaded on the third stream while the GPU works through
    # the seven sums over c (an upload only synchronises the stream it is issued on)
    if isinstance(h_coeffs, np.ndarray):
        h_arr = scalars_to_array(h_coeffs)
    else:
        h_arr = scalars_to_array([h_coeffs[i] for i in range(len(h))])
    h_head = h_ctx.upload(h_arr) if len(h_arr) else None
    pending_h = hv.launch(h_ctx, h_head, len(h_arr))
    ctx.sync()
    h_ctx.sync()
    raw = ctx.download(out_g1.ptr, 96 * len(g1)).tobytes() + h_ctx.download(pending_h[1].ptr, 96).tobytes()
    points = _from_jacobian_many_g1([raw[96 * i:96 * i + 96] for i in range(len(g1) + 1)])
    out = dict(zip(_SHARED_G1, points[:-1]))
    out[pending_twist[0]] = key.vectors[pending_twist[0]].result(pending_twist[1])
    out["h*g1"] = points[-1]
    return {name: out[name] for name in list(_ELEMENTS) + ["h*g1"]}


def _mid_is_identity(key, n_wires):
    flag = getattr(key, "_mid_identity", None)
    if flag is None:
        flag = key._mid_identity = bool(len(key.mid) and key.mid[0] == 0 and key.mid[-1] == len(key.mid) - 1
                                        and np.array_equal(key.mid_index, np.arange(len(key.mid))))
    return flag and n_wires == len(key.mid)


def compute_proof(qap, c, h, evalkey, deltas=None):
    """Pinocchio proof elements (pynocchio.py:228-273), one MSM per element.  `evalkey` is the
    reference's dict of points, or a PreparedKey made from it (device-resident, tabulated)."""
    if isinstance(evalkey, PreparedKey):
        return _compute_proof_prepared(evalkey, c, h, deltas)
    mid = list(qap.indices_mid)
    cm = [int(c[i]) for i in mid]

    def element(key_fmt, zk=()):
        scalars = list(cm) + [int(d) for d, _ in zk]
        points = [evalkey[key_fmt(i)] for i in mid] + [evalkey[name] for _, name in zk]
        return msm(scalars, points)

    zk = (lambda *pairs: pairs) if deltas is not None else (lambda *pairs: ())
    dv, dw, dy = (deltas.v, deltas.w, deltas.y) if deltas is not None else (0, 0, 0)
    h_scalars = [int(h.coeffs[i]) for i in range(0, len(h))]
    h_points = [evalkey["s^" + str(i) + "*g1"] for i in range(0, len(h))]
    return {
        "r_v*v_mid*g1": element(lambda i: "r_v*v" + str(i) + "*g1", zk((dv, "r_v*t*g1"))),
        "r_w*w_mid*g2": element(lambda i: "r_w*w" + str(i) + "*g2", zk((dw, "r_w*t*g2"))),
        "r_y*y_mid*g1": element(lambda i: "r_y*y" + str(i) + "*g1", zk((dy, "r_y*t*g1"))),
        "r_v*alpha_v*v_mid*g1": element(lambda i: f"r_v*alpha_v*v{i}*g1", zk((dv, "r_v*alpha_v*t*g1"))),
        "r_w*alpha_w*w_mid*g1": element(lambda i: f"r_w*alpha_w*w{i}*g1", zk((dw, "r_w*alpha_w*t*g1"))),
        "r_y*alpha_y*y_mid*g1": element(lambda i: f"r_y*alpha_y*y{i}*g1", zk((dy, "r_y*alpha_y*t*g1"))),
        "r_v*beta*v_mid+r_w*beta*w_mid+r_y*beta*y_mid*g1": element(
            lambda i: f"r_v*beta*v+r_w*beta*w+r_y*beta*y{i}_g1",
            zk((dv, "r_v*beta*t*g1"), (dw, "r_w*beta*t*g1"), (dy, "r_y*beta*t*g1"))),
        "h*g1": msm(h_scalars, h_points),
    }
