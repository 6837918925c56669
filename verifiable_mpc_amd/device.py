"""Device-resident vectors of the AC20 hot path: generators (PointVector) and witness /
linear-form coefficients (ScalarVector), plus the per-process GPU context.

Layout in HBM:
  PointVector.affine : n x 64 B   x||y, canonical          -> MSM / fold input
  PointVector.proj   : n x 96 B   X||Y||Z, representative   -> transcript text, exact fold
  ScalarVector.data  : n x 32 B   canonical residue mod l
Slices are zero-copy views (pointer + offset), which is what the halving rounds of
Protocol 4 use (verifiable_mpc/ac20/compressed_pivot.py:35-38).
"""
import os

import numpy as np

from . import _native, formats
from .groups import ORDER, Ed25519Point, as_point

_CTX = None


def get_context():
    """One vmpc context per process, on LOCAL_RANK's GPU (one process per GPU)."""
    global _CTX
    if _CTX is None:
        ndev, info = _native.backend_info()
        if ndev < 1:
            raise RuntimeError(f"no MI355X visible ({info}); the AC20 hot path has no CPU fallback")
        dev = int(os.environ.get("LOCAL_RANK", "0")) % ndev
        _CTX = _native.Context(dev)
    return _CTX


_AUX = {}


def get_aux_context(index=0):
    """Additional contexts (own stream + workspace) on the same GPU: index 0 runs B_i next to
    A_i (the two independent MSMs of a Protocol-4 round), index 1 the small Q' product."""
    if index not in _AUX:
        _AUX[index] = _native.Context(get_context().device)
    return _AUX[index]


def reset_context():
    global _CTX
    for c in list(_AUX.values()) + [_CTX]:
        if c is not None:
            c.close()
    _AUX.clear()
    _CTX = None


def reduce_scalar(v):
    """Canonical residue of an int / field element (negative and oversized ints allowed:
    every base has order l, SURVEY.md hard part 5)."""
    return int(v) % ORDER


class _View:
    """(buffer, element offset, length) with a fixed element stride."""
    __slots__ = ("buf", "off", "n", "stride")

    def __init__(self, buf, off, n, stride):
        self.buf, self.off, self.n, self.stride = buf, off, n, stride

    @property
    def ptr(self):
        return self.buf.ptr + self.off * self.stride

    def sub(self, a, b):
        return _View(self.buf, self.off + a, b - a, self.stride)


def _slice_bounds(key, n):
    a, b, step = key.indices(n)
    if step != 1:
        raise ValueError("device vectors only support contiguous slices")
    return a, max(a, b)


class DeviceScalar:
    """One residue mod l in device memory (e.g. an inner product nobody on the host needs to see)."""

    def __init__(self, buf, ctx):
        self.buf, self.ctx = buf, ctx

    @property
    def ptr(self):
        return self.buf.ptr

    def __int__(self):
        return int.from_bytes(self.ctx.download(self.buf.ptr, 32).tobytes(), "little")


class ScalarVector:
    """n scalars mod l on the device."""

    def __init__(self, view, ctx=None):
        self.ctx = ctx or get_context()
        self.v = view

    @classmethod
    def from_ints(cls, values, ctx=None):
        ctx = ctx or get_context()
        arr = _native.ints_to_array([reduce_scalar(v) for v in values], 32)
        return cls(_View(ctx.upload(arr), 0, len(values), 32), ctx)

    @classmethod
    def from_array(cls, arr, ctx=None):
        """(n, 32) uint8 canonical little-endian residues."""
        ctx = ctx or get_context()
        arr = _native.as_bytes_array(arr, 32)
        return cls(_View(ctx.upload(arr), 0, len(arr), 32), ctx)

    @classmethod
    def empty(cls, n, ctx=None):
        ctx = ctx or get_context()
        return cls(_View(ctx.alloc(max(1, 32 * n)), 0, n, 32), ctx)

    def __len__(self):
        return self.v.n

    @property
    def ptr(self):
        return self.v.ptr

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b = _slice_bounds(key, self.v.n)
            return ScalarVector(self.v.sub(a, b), self.ctx)
        if key < 0:
            key += self.v.n
        raw = self.ctx.download(self.v.ptr + 32 * key, 32)
        return int.from_bytes(raw.tobytes(), "little")

    def to_ints(self):
        return _native.array_to_ints(self.ctx.download(self.v.ptr, 32 * self.v.n, (self.v.n, 32)))

    def concat(self, values):
        """self + [v, ...] as a new vector (z_hat = z + [phi], compressed_pivot.py:136)."""
        extra = _native.ints_to_array([reduce_scalar(v) for v in values], 32)
        out = ScalarVector.empty(self.v.n + len(values), self.ctx)
        self.ctx.copy(out.ptr, self.ptr, 32 * self.v.n)
        self.ctx.upload_into(out.ptr + 32 * self.v.n, extra)
        return out

    def __add__(self, other):
        if isinstance(other, (list, tuple)):
            return self.concat(other)
        return NotImplemented

    # ---- arithmetic (csrc/frvec.hip) ------------------------------------------------------
    def axpy(self, c, y):
        """c * self + y  (element-wise, mod l)."""
        assert len(y) == len(self)
        out = ScalarVector.empty(len(self), self.ctx)
        self.ctx.fr_axpy(reduce_scalar(c), self.ptr, y.ptr, len(self), out.ptr)
        return out

    def axpy_concat(self, c, y, tail):
        """(c * self + y) + [tail]  (y None: c * self) in one pass: z_hat = (c0 x + r) || phi,
        L~ = (L || 0) * c1 (compressed_pivot.py:134-141)"""
        assert y is None or len(y) == len(self)
        out = ScalarVector.empty(len(self) + 1, self.ctx)
        self.ctx.fr_axpy_tail(reduce_scalar(c), self.ptr, y.ptr if y is not None else None, len(self),
                              reduce_scalar(tail), out.ptr)
        return out

    def scale(self, c):
        out = ScalarVector.empty(len(self), self.ctx)
        self.ctx.fr_scale(reduce_scalar(c), self.ptr, len(self), out.ptr)
        return out

    def dot(self, other):
        assert len(other) == len(self)
        return self.ctx.fr_dot(self.ptr, other.ptr, len(self))

    def dot_dev(self, other):
        """<self, other> mod l as a DeviceScalar: stays on the device, nothing waits for it"""
        assert len(other) == len(self)
        return DeviceScalar(self.ctx.fr_dot_to_dev(self.ptr, other.ptr, len(self)), self.ctx)

    def text_begin(self, is_signed=None):
        """start producing the transcript text on the side stream (no host wait): formatting and
        the device->host copy then overlap the kernels that follow on the main stream"""
        if is_signed is None:
            is_signed = formats.scalar_signed()
        if len(self):
            side = get_aux_context(2)
            side.wait_for(self.ctx)
            self._pending_text = (is_signed, side.format_begin("scalars", self.ptr, len(self), is_signed,
                                                                    keepalive=self.v.buf))

    def text(self, is_signed=None):
        """b'v0, v1, ..., ' as produced on the device (uint8 array)."""
        if is_signed is None:
            is_signed = formats.scalar_signed()
        pend = getattr(self, "_pending_text", None)
        if pend is not None and pend[0] == is_signed:
            return pend[1].result()
        return self.ctx.format_scalars(self.ptr, len(self), is_signed)

    def text_chunks(self, is_signed=None):
        """text()[:-2] in pieces, each as soon as it is on the host (a text_begin() in flight), else at once"""
        if is_signed is None:
            is_signed = formats.scalar_signed()
        pend = getattr(self, "_pending_text", None)
        if pend is not None and pend[0] == is_signed:
            return pend[1].chunks(trim=2)
        return iter([memoryview(self.text(is_signed))[:-2]])

    def __repr__(self):
        body = self.text().tobytes().decode()
        return "[" + body[:-2] + "]"


# Measured at 2^20 generators with the round-context prover (5 rounds + one multi-round fold on this table, the
# rest on the folded vector's own): 24.2 / 21.3 / 20.7 / 20.0 / 21.0 ms per proof for 1 / 2 / 4 / 8 / 16 rows.
TABLE_BUDGET_BYTES = 1 << 30


class FixedBaseTable:
    """Device table 2^(16 w) * P_i over a generator vector followed by a few extra base points."""

    def __init__(self, buf, n, extra_bytes, rows):
        self.buf, self.n, self.extra_bytes, self.rows = buf, n, extra_bytes, rows

    @property
    def ptr(self):
        return self.buf.ptr

    def extra_index(self, point):
        """position of `point` among the extras, or None"""
        try:
            return self.extra_bytes.index(as_point(point).to_affine_bytes())
        except ValueError:
            return None


class PointVector:
    """n Ed25519 points on the device, affine always, projective representatives when the
    reference transcript needs them (`keep_proj`)."""

    def __init__(self, affine_view, proj_view=None, ctx=None):
        self.ctx = ctx or get_context()
        self._n = 0
        self._a_make = None      # deferred affine array (concat of a tabulated vector: built if anybody reads it)
        self._p_make = None      # ... and the projective representatives, likewise
        self.a = affine_view
        self.p = proj_view
        self._digest = None
        self._table = None       # FixedBaseTable over this vector (precompute)
        self._table_tail = 0     # trailing elements that are the table's extras 0.._table_tail-1
        self._wide = None        # the 13-row wide-window table over the same vector and extras (precompute(wide=True))

    # ---- construction ----------------------------------------------------------------------
    @classmethod
    def from_points(cls, points, ctx=None, keep_proj=True):
        """From host elements - ours or foreign three-coordinate projective ones, e.g. the list of MPyC points a
        reference caller holds (representatives preserved)."""
        ctx = ctx or get_context()
        points = [as_point(p) for p in points]
        n = len(points)
        proj = np.frombuffer(b"".join(p.to_proj_bytes() for p in points), dtype=np.uint8)
        pbuf = ctx.upload(proj.reshape(n, 96) if n else np.zeros((0, 96), np.uint8))
        abuf = ctx.alloc(max(1, 64 * n))
        ctx.normalize(pbuf.ptr, n, abuf.ptr)
        return cls(_View(abuf, 0, n, 64), _View(pbuf, 0, n, 96) if keep_proj else None, ctx)

    @classmethod
    def from_affine_array(cls, arr, ctx=None, keep_proj=False, validate=True):
        """(n, 64) uint8 x||y canonical little-endian."""
        ctx = ctx or get_context()
        arr = _native.as_bytes_array(arr, 64)
        n = len(arr)
        abuf = ctx.upload(arr)
        if validate and ctx.validate_points(abuf.ptr, n):
            raise _native.VmpcError(_native.E_NOTONCURVE, "PointVector.from_affine_array")
        pv = None
        if keep_proj:
            pbuf = ctx.alloc(max(1, 96 * n))
            ctx.affine_to_proj(abuf.ptr, n, pbuf.ptr)
            pv = _View(pbuf, 0, n, 96)
        return cls(_View(abuf, 0, n, 64), pv, ctx)

    @classmethod
    def fixed_base(cls, base, exponents, ctx=None, keep_proj=True):
        """[base ** r for r in exponents] (circuit_sat_r1cs.py:64-70) on the device;
        `exponents` is a ScalarVector or a list of ints."""
        ctx = ctx or get_context()
        if not isinstance(exponents, ScalarVector):
            exponents = ScalarVector.from_ints(exponents, ctx)
        n = len(exponents)
        if not keep_proj:
            # only the group elements are wanted: comb table of the one base (vmpc_fixed_base_dev)
            bbuf = ctx.upload(np.frombuffer(base.to_affine_bytes(), dtype=np.uint8))
            abuf = ctx.alloc(max(1, 64 * n))
            ctx.fixed_base(bbuf.ptr, exponents.ptr, n, abuf.ptr)
            ctx.sync()
            return cls(_View(abuf, 0, n, 64), None, ctx)
        bbuf = ctx.upload(np.frombuffer(base.to_proj_bytes(), dtype=np.uint8))
        abuf = ctx.alloc(max(1, 64 * n))
        pbuf = ctx.alloc(max(1, 96 * n)) if keep_proj else None
        ctx.repeat(bbuf.ptr, 1, False, exponents.ptr, n, False, pbuf.ptr if pbuf else None, abuf.ptr)
        ctx.sync()
        return cls(_View(abuf, 0, n, 64), _View(pbuf, 0, n, 96) if keep_proj else None, ctx)

    # ---- list protocol ----------------------------------------------------------------------
    def __len__(self):
        return self._n

    @property
    def a(self):
        if self._a is None and self._a_make is not None:
            self._a, self._a_make = self._a_make(), None
        return self._a

    @a.setter
    def a(self, view):
        self._a = view
        if view is not None:
            self._n = view.n

    @property
    def affine_ptr(self):
        return self.a.ptr

    @property
    def proj_ptr(self):
        return self.p.ptr if self.has_proj else None

    @property
    def has_proj(self):
        return self._p is not None or self._p_make is not None

    @property
    def p(self):
        if self._p is None and self._p_make is not None:
            self._p, self._p_make = self._p_make(), None
        return self._p

    @p.setter
    def p(self, view):
        self._p = view

    def __getitem__(self, key):
        if isinstance(key, slice):
            a, b = _slice_bounds(key, self.a.n)
            sub = PointVector(self.a.sub(a, b), self.p.sub(a, b) if self.has_proj else None,
                              self.ctx)
            if a == 0 and self._table is not None:
                sub._table = self._table       # a prefix addresses the same table rows
                sub._wide = self._wide
                sub._table_tail = max(0, b - (len(self) - self._table_tail))
            return sub
        if key < 0:
            key += self.a.n
        if self.has_proj:
            return Ed25519Point.from_proj_bytes(self.ctx.download(self.p.ptr + 96 * key, 96).tobytes())
        return Ed25519Point.from_affine_bytes(self.ctx.download(self.a.ptr + 64 * key, 64).tobytes())

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def to_points(self):
        n = len(self)
        if self.has_proj:
            raw = self.ctx.download(self.p.ptr, 96 * n).tobytes()
            return [Ed25519Point.from_proj_bytes(raw[96 * i:96 * i + 96]) for i in range(n)]
        raw = self.ctx.download(self.a.ptr, 64 * n).tobytes()
        return [Ed25519Point.from_affine_bytes(raw[64 * i:64 * i + 64]) for i in range(n)]

    def affine_array(self):
        return self.ctx.download(self.a.ptr, 64 * len(self), (len(self), 64))

    def concat(self, points):
        """self + [pt, ...] as a new vector (g_hat = g + [h], compressed_pivot.py:138)."""
        points = [as_point(p) for p in points]
        n, m = len(self), len(points)
        t = self._table
        tabulated = t is not None and self._table_tail == 0 and n == t.n and \
            [p.to_affine_bytes() for p in points] == t.extra_bytes[:m]
        src, ctx = self, self.ctx

        def affine():
            abuf = ctx.alloc(64 * (n + m))
            ctx.copy(abuf.ptr, src.a.ptr, 64 * n)
            ctx.upload_into(abuf.ptr + 64 * n,
                            np.frombuffer(b"".join(p.to_affine_bytes() for p in points), np.uint8))
            return _View(abuf, 0, n + m, 64)

        def proj():
            pbuf = ctx.alloc(96 * (n + m))
            ctx.copy(pbuf.ptr, src.p.ptr, 96 * n)
            ctx.upload_into(pbuf.ptr + 96 * n,
                            np.frombuffer(b"".join(p.to_proj_bytes() for p in points), np.uint8))
            return _View(pbuf, 0, n + m, 96)

        if tabulated:
            # g + [h] with h an extra of g's table: commitments over the result run on the table, so the copies of
            # 2^20 generators (64 MB affine, 96 MB projective, per proof) wait until somebody reads the arrays
            out = PointVector(None, None, self.ctx)
            out._n, out._a_make = n + m, affine
            if self.has_proj:
                out._p_make = proj
        else:
            out = PointVector(affine(), proj() if self.has_proj else None, self.ctx)
        if tabulated:
            out._table, out._table_tail = t, m      # g + [h]: h is extra 0 of g's table
            out._wide = self._wide
        out._text_parent = (src, list(points))      # its transcript text = the source's text + these points' (text_chunks)
        return out

    def __add__(self, other):
        if isinstance(other, (list, tuple)):
            return self.concat(list(other))
        return NotImplemented

    # ---- kernels -------------------------------------------------------------------------------
    def precompute(self, extras=(), rows=None, wide=False):
        """Build the fixed-base table (include/vmpc.h: vmpc_msm_table_build_dev) for this vector and
        the `extras` (the commitment bases h, k of the CRS).  Commitments over this vector or a
        prefix of it, with one of `extras` as base point, then need no point preparation and only
        (16/rows - 1) * 16 doublings of window recombination.  `rows` in {1, 2, 4, 8, 16} (128 bytes
        of HBM per generator and row); by default the largest with a table of at most 1 GiB: the
        bucket stage gathers table entries at random, and far past the Infinity Cache that costs what
        the shorter recombination saves (at 2^20 generators every choice is within 3 % for one
        commitment; 8 rows is the best for the prover, see TABLE_BUDGET_BYTES).
        rows = 13 is the WIDE-WINDOW table (round 6): rows spaced 20 bits, a commitment is 13 mixed additions per
        term into one set of 2^19 buckets (instead of 16 into 16 / rows sets of 2^15) and needs no recombination -
        the fastest form for plain commitments over a large CRS (1.7 GB at 2^20 generators); the prover's round
        context and fold jump do not read it (compressed_pivot._tabulated)."""
        extras = [as_point(p) for p in extras]
        if rows is None and os.environ.get("VMPC_TABLE_ROWS"):
            rows = int(os.environ["VMPC_TABLE_ROWS"])          # tuning knob
        if rows is None:
            rows = 16
            while rows > 1 and rows * 128 * len(self) > TABLE_BUDGET_BYTES:
                rows //= 2
        raw = b"".join(p.to_affine_bytes() for p in extras)
        eb = self.ctx.upload(np.frombuffer(raw, np.uint8)) if extras else None
        buf = self.ctx.msm_table_build(self.a.ptr, len(self), eb.ptr if eb else None, len(extras), rows)
        self.ctx.sync()
        self._table = FixedBaseTable(buf, len(self), [raw[64 * i:64 * i + 64] for i in range(len(extras))], rows)
        self._wide = None
        if wide and rows != 13 and len(self) + len(extras) <= (1 << 22) - 8192:
            # BESIDE it, the 13-row wide-window table over the same columns: commitments (pivot.vector_commitment, the
            # prover's A and the A_i, B_i of its rounds before the fold) read this one - 13 mixed additions per term -
            # while the fold jump keeps the table above, whose rows are spaced 256 / rows bits
            wbuf = self.ctx.msm_table_build(self.a.ptr, len(self), eb.ptr if eb else None, len(extras), 13)
            self.ctx.sync()
            self._wide = FixedBaseTable(wbuf, len(self), list(self._table.extra_bytes), 13)
        return self

    TEXT_SLICE = 1 << 16     # fold(stream_text=True): elements folded, formatted and sent to the host at a time
    TEXT_FIRST_SLICE = int(os.environ.get("VMPC_FOLD_FIRST_SLICE", str(1 << 13)))      # ... the first time
    TEXT_SLICED_FROM = int(os.environ.get("VMPC_FOLD_SLICED_FROM", str(1 << 14)))      # shorter folds are not sliced

    def fold(self, other, c, keep_proj=None, stream_text=False, after_first=None):
        """[(self[i] ** c) * other[i]] (compressed_pivot.py:64/:178), csrc/exact.hip k_fold.
        stream_text: the result's transcript text is wanted next (the reference transcript hashes the folded
        generators every round, compressed_pivot.py:52): a long vector is folded slice by slice, each slice formatted
        and copied on the side stream as soon as it exists, so the host hashes the first slices while the rest is
        still being folded (an exact 2^19-element fold is 8.6 ms; hashing its 123 MB of text takes 50).
        after_first(more): called once the first slice (the whole fold, if it is not sliced) and its text are
        enqueued - the caller's own work on ANOTHER stream that should run beside that slice and ahead of the rest;
        more: further slices will be enqueued on this vector's stream after the call returns."""
        assert len(self) == len(other)
        half = len(self)
        if keep_proj is None:
            keep_proj = self.has_proj
        abuf = self.ctx.alloc(max(1, 64 * half))
        pbuf = self.ctx.alloc(max(1, 96 * half)) if keep_proj else None
        use_proj = self.has_proj and other.has_proj
        lp, rp = (self.p.ptr, other.p.ptr) if use_proj else (self.a.ptr, other.a.ptr)
        stride = 96 if use_proj else 64
        out = PointVector(_View(abuf, 0, half, 64), _View(pbuf, 0, half, 96) if pbuf else None, self.ctx)
        # (with a co-runner - after_first: the prover's pair on another stream - the plan of round 6's first half stays:
        # folds of 2^17 elements or more, a first slice of 2^14, then full slices; the growing plan measured 31 against
        # 28 ms outside the hash there: the formatter's kernels do not get in beside the pair's bucket kernel AND the
        # two-wave fold, and the first text is late by the pair)
        sliced_from = 2 * self.TEXT_SLICE if after_first is not None else self.TEXT_SLICED_FROM
        if not (stream_text and pbuf is not None and half >= sliced_from):
            self.ctx.fold(lp, rp, not use_proj, reduce_scalar(c), half, pbuf.ptr if pbuf else None, abuf.ptr)
            if stream_text:
                out.text_begin()
            if after_first is not None:
                after_first(False)
            return out
        side, pieces = get_aux_context(2), []
        # The slices grow: the hash waits for the FIRST one's text and an exact fold is a 253-step ladder whatever the
        # slice's length - 0.5-0.6 ms on the two-wave kernel (<= 2^13 elements), 0.8 on the quad kernel (<= 2^14), 1.2-1.4
        # with one lane per element (csrc/exact.hip) - while hashing a slice's text takes 0.1 ms per 2^10 elements: each
        # slice is back before the host has hashed the one before it.
        cuts, step = [0], self.TEXT_FIRST_SLICE
        if after_first is not None:
            step = self.TEXT_SLICE // 4
        while cuts[-1] + step < half:
            cuts.append(cuts[-1] + step)
            step = min(2 * step, self.TEXT_SLICE) if after_first is None else self.TEXT_SLICE
        for a, b in zip(cuts, cuts[1:] + [half]):
            cnt = b - a
            self.ctx.fold(lp + stride * a, rp + stride * a, not use_proj, reduce_scalar(c), cnt, pbuf.ptr + 96 * a,
                          abuf.ptr + 64 * a)
            side.wait_for(self.ctx)
            pieces.append(side.format_begin("points", pbuf.ptr + 96 * a, cnt, keepalive=pbuf, own_signal=True))
            if a == 0 and after_first is not None:
                after_first(True)
        out._pending_text = (formats.point_style(), _native.TextSequence(pieces))
        return out

    def text_begin(self):
        """start producing the transcript text on the side stream (no host wait)"""
        if self.has_proj and len(self):
            side = get_aux_context(2)
            side.wait_for(self.ctx)
            # (remembered WITH the point format it was produced in: formats.set_reference_format takes effect at once)
            self._pending_text = (formats.point_style(), side.format_begin("points", self.p.ptr, len(self),
                                                                           keepalive=self.p.buf))

    def text(self):
        """b'[X, Y, Z], [X, Y, Z], ..., ' (uint8 array) for the Fiat-Shamir pre-image."""
        if not self.has_proj:
            raise ValueError("projective representatives were not kept for this vector")
        pend = getattr(self, "_pending_text", None)
        if pend is not None and pend[0] == formats.point_style():
            return pend[1].result()
        return self.ctx.format_points(self.p.ptr, len(self))

    def text_chunks(self):
        """text()[:-2] in pieces, each as soon as it is on the host (a text_begin() in flight), else at once"""
        if not self.has_proj:
            raise ValueError("projective representatives were not kept for this vector")
        pend = getattr(self, "_pending_text", None)
        if pend is not None and pend[0] == formats.point_style():
            return pend[1].chunks(trim=2)
        parent = getattr(self, "_text_parent", None)
        if parent is not None:
            # g_hat = g + [h] right after g itself went into a hash (compressed_pivot.py:125-138): g's text is on
            # the host already (or on its way) - 246 MB at 2^20 points that need not be formatted and copied again
            src, pts = parent
            ppend = getattr(src, "_pending_text", None)
            if ppend is not None and ppend[0] == formats.point_style():
                def pieces():
                    yield from ppend[1].chunks(trim=0)
                    tail = PointVector.from_points(pts, self.ctx).text()
                    yield memoryview(tail)[:-2]
                return pieces()
        return iter([memoryview(self.text())[:-2]])

    def __repr__(self):
        body = self.text().tobytes().decode()
        return "[" + body[:-2] + "]"
