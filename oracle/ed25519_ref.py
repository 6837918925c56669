"""ORACLE (test infrastructure, NOT the product): Ed25519 group layer, big-int Python.

This file is a CPU restatement of the group arithmetic that the reference's hot
path reaches through MPyC operators (`**`, `*`, `==`, `.normalize()`, `repr`):
    verifiable_mpc/ac20/pivot.py:26-28,143-144,170,187
    verifiable_mpc/ac20/compressed_pivot.py:41-42,52,64,66,118,140,178,180,193
    verifiable_mpc/ac20/circuit_sat_r1cs.py:62-70,81
    demos/demo_zkp_ac20.py:46-49  (EllipticCurve('Ed25519','projective'))

The arithmetic itself lives in the third-party dependency `mpyc` (requirement
`mpyc >= 0.8`, unpinned: /root/reference/setup.py:28), which is NOT vendored under
/root/reference and cannot be installed here.  What is restated below is therefore
MPyC's *published* algorithm as recalled [mpyc-recall]:
  * points are projective (X:Y:Z) on -x^2 + y^2 = 1 + d x^2 y^2 over GF(2^255-19);
  * group operation  = EFD add-2008-bbjlp  (twisted Edwards, projective);
  * doubling         = EFD dbl-2008-bbjlp;
  * a ** n           = right-to-left binary double-and-add (`repeat`), negative n
                       inverts the base first;
  * mpctools.reduce  = balanced pairwise tree, `initial` appended at the END;
  * repr(point)      = repr of the list of its three coordinates, each coordinate
                       printed as an unsigned decimal integer;
  * repr(scalar in GF(l)) = SIGNED decimal integer in (-l/2, l/2].

PARITY STATUS: the curve layer is pinned by RFC 8032 constants/KATs and by 256 OpenSSL-made key pairs and
signatures (tests/test_oracle_golden.py, tests/test_oracle_openssl.py); the byte-level formats of real MPyC (repr
brackets, signedness, exact projective formulas) are "parity unpinned" - they
cannot be checked in this container.  Group elements are canonical once
affine-normalised, so affine-level results do not depend on that recall.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""

P = 2**255 - 19
ELL = 2**252 + 27742317777372353535851937790883648493  # prime subgroup order
D = (-121665 * pow(121666, P - 2, P)) % P
D2 = (2 * D) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)

# representation switches of the [mpyc-recall] formats (one place to change)
POINT_REPR_OPEN = "["
POINT_REPR_CLOSE = "]"
COORD_REPR_SIGNED = False       # a coordinate c > (p - 1) / 2 printed as -(p - c)
SCALAR_REPR_SIGNED = True


def set_format(point_brackets="[]", coord_signed=False, scalar_signed=True):
    """the oracle's side of verifiable_mpc_amd.set_reference_format (same arguments, same meaning)"""
    global POINT_REPR_OPEN, POINT_REPR_CLOSE, COORD_REPR_SIGNED, SCALAR_REPR_SIGNED
    assert point_brackets in ("[]", "()")
    POINT_REPR_OPEN, POINT_REPR_CLOSE = point_brackets[0], point_brackets[1]
    COORD_REPR_SIGNED, SCALAR_REPR_SIGNED = bool(coord_signed), bool(scalar_signed)


def _recover_x(y, sign):
    """RFC 8032 section 5.1.3 x-recovery."""
    y2 = y * y % P
    u = (y2 - 1) % P
    v = (D * y2 + 1) % P
    x2 = u * pow(v, P - 2, P) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P != 0:
        x = x * SQRT_M1 % P
    if (x * x - x2) % P != 0:
        raise ValueError("not a square")
    if (x & 1) != sign:
        x = P - x
    return x


BASE_Y = 4 * pow(5, P - 2, P) % P
BASE_X = _recover_x(BASE_Y, 0)
BASE = (BASE_X, BASE_Y, 1)      # group.generator (circuit_sat_r1cs.py:62)
IDENTITY = (0, 1, 1)            # group.identity  (pivot.py:28)


def on_curve(pt):
    x, y, z = pt
    zi = pow(z, P - 2, P)
    x, y = x * zi % P, y * zi % P
    return (-x * x + y * y - 1 - D * x * x % P * y * y) % P == 0


def pt_add(p1, p2):
    """Group operation, EFD add-2008-bbjlp with a = -1 [mpyc-recall].

    Restates what `a * b` does for two points (compressed_pivot.py:64,66).
    """
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    a = z1 * z2 % P
    b = a * a % P
    c = x1 * x2 % P
    d = y1 * y2 % P
    e = D * c % P * d % P
    f = (b - e) % P
    g = (b + e) % P
    x3 = a * f % P * (((x1 + y1) * (x2 + y2) - c - d) % P) % P
    y3 = a * g % P * ((d + c) % P) % P        # D - a*C with a = -1
    z3 = f * g % P
    return (x3, y3, z3)


def pt_dbl(p1):
    """Doubling, EFD dbl-2008-bbjlp with a = -1 [mpyc-recall]."""
    x1, y1, z1 = p1
    b = (x1 + y1) * (x1 + y1) % P
    c = x1 * x1 % P
    d = y1 * y1 % P
    e = (-c) % P
    f = (e + d) % P
    h = z1 * z1 % P
    j = (f - 2 * h) % P
    x3 = (b - c - d) * j % P
    y3 = f * ((e - d) % P) % P
    z3 = f * j % P
    return (x3, y3, z3)


def pt_neg(p1):
    x, y, z = p1
    return ((-x) % P, y, z)


def pt_repeat(a, n):
    """`a ** n` (pivot.py:143, compressed_pivot.py:64): right-to-left binary
    double-and-add; `n` may be negative or larger than the group order
    (compressed_pivot.py:66 uses c**2 unreduced) [mpyc-recall]."""
    if n == 0:
        return IDENTITY
    if n < 0:
        a = pt_neg(a)
        n = -n
    d = a
    c = IDENTITY
    for i in range(n.bit_length() - 1):
        if (n >> i) & 1:
            c = pt_add(c, d)
        d = pt_dbl(d)
    return pt_add(c, d)


def tree_reduce(f, xs, initial=None):
    """mpctools.reduce as used by pivot.list_mul (pivot.py:26-28): pairwise
    balanced tree; the initial value is appended at the end [mpyc-recall]."""
    xs = list(xs)
    if initial is not None:
        xs.append(initial)
    if not xs:
        raise TypeError("reduce() of empty sequence with no initial value")
    while len(xs) > 1:
        odd = len(xs) % 2
        xs[odd:] = [f(xs[i], xs[i + 1]) for i in range(odd, len(xs), 2)]
    return xs[0]


def pt_normalize(p1):
    """`.normalize()` (pivot.py:170, compressed_pivot.py:52,118): scale to Z = 1."""
    x, y, z = p1
    zi = pow(z, P - 2, P)
    return (x * zi % P, y * zi % P, 1)


def pt_affine(p1):
    x, y, _ = pt_normalize(p1)
    return (x, y)


def pt_eq(p1, p2):
    """Projective-aware `==` (compressed_pivot.py:197)."""
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    return (x1 * z2 - x2 * z1) % P == 0 and (y1 * z2 - y2 * z1) % P == 0


def pt_repr(p1):
    """repr() of a point as it enters str(input_list) in pivot.py:134 [mpyc-recall]."""
    def c(v):
        v %= P
        return v - P if COORD_REPR_SIGNED and v > P // 2 else v
    x, y, z = p1
    return f"{POINT_REPR_OPEN}{c(x)}, {c(y)}, {c(z)}{POINT_REPR_CLOSE}"


def scalar_repr(v):
    """repr() of a GF(l) element (signed residue) [mpyc-recall]."""
    v %= ELL
    if SCALAR_REPR_SIGNED and v > ELL // 2:
        v -= ELL
    return str(v)


def scalar_int(v):
    """int() of a GF(l) element as pivot._int sees it (pivot.py:119-128)."""
    v %= ELL
    if SCALAR_REPR_SIGNED and v > ELL // 2:
        v -= ELL
    return v


# ---- RFC 8032 encoding (used only to pin the curve layer with KATs) ----------

def encode_rfc8032(p1):
    x, y = pt_affine(p1)
    return int.to_bytes(y | ((x & 1) << 255), 32, "little")


def decode_rfc8032(b):
    v = int.from_bytes(b, "little")
    sign = v >> 255
    y = v & ((1 << 255) - 1)
    if y >= P:
        raise ValueError("non-canonical y")
    x = _recover_x(y, sign)
    return (x, y, 1)


def rfc8032_public_key(seed32):
    import hashlib
    h = hashlib.sha512(seed32).digest()
    a = int.from_bytes(h[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    return encode_rfc8032(pt_repeat(BASE, a))


# ---- byte formats shared with the C-ABI (include/vmpc.h) ---------------------

def affine_to_bytes(p1):
    """64 bytes x||y little-endian, canonical residues."""
    x, y = pt_affine(p1)
    return x.to_bytes(32, "little") + y.to_bytes(32, "little")


def affine_from_bytes(b):
    return (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little"), 1)


def proj_to_bytes(p1):
    """96 bytes X||Y||Z little-endian, canonical residues, representative kept."""
    return b"".join(int(c % P).to_bytes(32, "little") for c in p1)


def proj_from_bytes(b):
    return tuple(int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(3))
