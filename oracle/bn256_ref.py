"""ORACLE (test infrastructure, NOT the product): BN-256 G1 / G2 multi-scalar products as the
reference's Pinocchio prover computes them (BASELINE config 5).

Restates, with Python big ints,
    verifiable_mpc/trinocchio/pynocchio.py:228-246  compute_proof: eight sums
        [int(c[i]) * evalkey[...] for i in qap.indices_mid]  folded by apply_to_list(point_add, ..)
    verifiable_mpc/trinocchio/pynocchio.py:82-93      apply_to_list: recursive halving tree
    verifiable_mpc/ac20/pairing.py:44-51              curve parameters (v = 1868033, u = v^3)
on the groups the reference obtains from MPyC: EllipticCurve('BN256', 'jacobian') and
EllipticCurve('BN256_twist', 'jacobian') (demos/demo_zkp_pynocchio.py:27-30).

MPyC is not vendored; what is restated is the published curve (Barreto-Naehrig, y^2 = x^3 + 3
over F_p, sextic twist y^2 = x^3 + 3/xi over F_p[i]/(i^2+1), xi = i + 3 - the parameters of
randombit/pairings.py which the reference credits, pairing.py:1-40) [mpyc-recall for the choice of
generators: G1 = (1, -2); the twist generator below is the pairings.py / golang bn256 one].
Only AFFINE results are compared (proof elements are consumed by pairings, which do not depend
on the Jacobian representative), so the exact Jacobian formulas MPyC uses do not matter here.

PARITY STATUS: curve constants pinned by internal known answers (n*G = O on both groups, points on
curve, p and n from v); protocol shape pinned by tests/golden/pynocchio_bn256.json (reference's
compute_proof run over the mpyc shim).  Real-MPyC byte formats: parity unpinned.
"""

V = 1868033
U = V ** 3
P = 36 * U ** 4 + 36 * U ** 3 + 24 * U ** 2 + 6 * U + 1
N = 36 * U ** 4 + 36 * U ** 3 + 18 * U ** 2 + 6 * U + 1      # group order (pairing.py:49-51)
B = 3


# ---- F_p and F_p2 = F_p[i]/(i^2 + 1); an F_p2 element is (real, imag) --------------------------

class Fp:
    zero, one = 0, 1

    @staticmethod
    def add(a, b): return (a + b) % P
    @staticmethod
    def sub(a, b): return (a - b) % P
    @staticmethod
    def mul(a, b): return a * b % P
    @staticmethod
    def neg(a): return (-a) % P
    @staticmethod
    def inv(a): return pow(a, P - 2, P)
    @staticmethod
    def is_zero(a): return a % P == 0
    @staticmethod
    def small(k): return k % P


class Fp2:
    zero, one = (0, 0), (1, 0)

    @staticmethod
    def add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    @staticmethod
    def sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
    @staticmethod
    def mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    @staticmethod
    def neg(a): return ((-a[0]) % P, (-a[1]) % P)
    @staticmethod
    def inv(a):
        d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
        return (a[0] * d % P, (-a[1] * d) % P)
    @staticmethod
    def is_zero(a): return a[0] % P == 0 and a[1] % P == 0
    @staticmethod
    def small(k): return (k % P, 0)


XI = (3, 1)
B_TWIST = Fp2.mul((3, 0), Fp2.inv(XI))

G1 = (1, P - 2)                                     # [mpyc-recall] (1, -2)
G2 = ((64746500191241794695844075326670126197795977525365406531717464316923369116492,
       21167961636542580255011770066570541300993051739349375019639421053990175267184),
      (17778617556404439934652658462602675281523610326338642107814333856843981424549,
       20666913350058776956210519119118544732556678129809273996262322366050359951122))


class Curve:
    """Short Weierstrass y^2 = x^3 + b, points affine (x, y) or None for infinity."""

    def __init__(self, F, b):
        self.F, self.b = F, b

    def on_curve(self, pt):
        if pt is None:
            return True
        F = self.F
        x, y = pt
        return F.is_zero(F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), self.b)))

    def neg(self, pt):
        return None if pt is None else (pt[0], self.F.neg(pt[1]))

    def add(self, p1, p2):
        F = self.F
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        x1, y1 = p1
        x2, y2 = p2
        if F.is_zero(F.sub(x1, x2)):
            if F.is_zero(F.add(y1, y2)):
                return None
            lam = F.mul(F.mul(F.small(3), F.mul(x1, x1)), F.inv(F.mul(F.small(2), y1)))
        else:
            lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    def mul(self, k, pt):
        """`int * point` (pynocchio.py:229): any integer, negative allowed."""
        if k < 0:
            return self.mul(-k, self.neg(pt))
        acc, d = None, pt
        while k:
            if k & 1:
                acc = self.add(acc, d)
            d = self.add(d, d)
            k >>= 1
        return acc

    def apply_to_list(self, pts):
        """pynocchio.py:82-93: recursive halving tree of point additions."""
        n = len(pts)
        if n == 1:
            return pts[0]
        return self.add(self.apply_to_list(pts[: n // 2]), self.apply_to_list(pts[n // 2:]))

    def msm(self, scalars, pts):
        """apply_to_list(point_add, [int(c_i) * P_i ...]) (pynocchio.py:229-246)."""
        return self.apply_to_list([self.mul(int(k), p) for k, p in zip(scalars, pts)])


E1 = Curve(Fp, B)
E2 = Curve(Fp2, B_TWIST)


# ---- byte formats of include/vmpc.h (BN-256 section) ----------------------------------------------
# G1 affine: 64 B x || y little-endian canonical; infinity = 64 zero bytes ((0,0) is not on the curve)
# G2 affine: 128 B x.real || x.imag || y.real || y.imag; infinity = 128 zero bytes

def g1_to_bytes(pt):
    if pt is None:
        return bytes(64)
    return pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


def g1_from_bytes(b):
    x, y = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little")
    return None if x == 0 and y == 0 else (x, y)


def g2_to_bytes(pt):
    if pt is None:
        return bytes(128)
    (x0, x1), (y0, y1) = pt
    return b"".join(v.to_bytes(32, "little") for v in (x0, x1, y0, y1))


def g2_from_bytes(b):
    v = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))
