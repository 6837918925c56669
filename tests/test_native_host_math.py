"""CPU check of the device math headers (fe25519/ge25519/fr/fmt .h).

The headers are `__host__ __device__`; this test builds them with g++ into a small
harness (tests/native/host_math_test.cpp) and compares every operation with the
Python oracle on random and edge-case operands.  It does not replace the `-m gpu`
parity tests (which call the kernels through the C-ABI); it catches arithmetic bugs
before GPU time is spent.
"""
import os
import random
import subprocess

import pytest

from oracle import ed25519_ref as ed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "host_math_test.cpp")
P, ELL = ed.P, ed.ELL


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    # AddressSanitizer + UndefinedBehaviorSanitizer on the CPU build of the device headers (SURVEY.md
    # section 5: sanitizers run here; GPU ASan is not available on the pool).  Unsigned wrap-around is
    # intended arithmetic in the limb code and is not part of -fsanitize=undefined.
    exe = str(tmp_path_factory.mktemp("native") / "host_math_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-o", exe, SRC])

    def run(lines):
        res = subprocess.run([exe], input="\n".join(lines) + "\nquit\n", text=True, capture_output=True,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0",
                                      UBSAN_OPTIONS="print_stacktrace=1"))
        assert res.returncode == 0 and "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, \
            res.stderr[-2000:]
        out = res.stdout.strip().split("\n")
        assert len(out) == len(lines), (len(out), len(lines))
        return out
    return run


def hx(v):
    return format(v, "x")


EDGE = [0, 1, 2, 19, 38, P - 1, P, P + 1, 2 * P, 2 * P + 37, 2**255, 2**256 - 1, 2**256 - 38,
        2**255 - 20, 2**32 - 1, 2**224 - 1, (2**256 - 1) ^ (2**128)]


def test_field_ops(harness):
    rng = random.Random(1)
    vals = EDGE + [rng.getrandbits(256) for _ in range(60)]
    pairs = [(a, b) for a in EDGE for b in EDGE] + \
            [(rng.choice(vals), rng.choice(vals)) for _ in range(300)]
    lines, want = [], []
    for a, b in pairs:
        lines += [f"femul {hx(a)} {hx(b)}", f"feadd {hx(a)} {hx(b)}", f"fesub {hx(a)} {hx(b)}"]
        want += [hx(a * b % P), hx((a + b) % P), hx((a - b) % P)]
    for a in vals:
        lines += [f"fesqr {hx(a)}", f"fecanon {hx(a)}", f"femulu32 {hx(a)} {hx(rng.getrandbits(32))}"]
        s = int(lines[-1].split()[-1], 16)
        want += [hx(a * a % P), f"{hx(a % P)} {1 if a < P else 0}", hx(a * s % P)]
        if a % P:
            lines.append(f"feinv {hx(a)}")
            want.append(hx(pow(a, P - 2, P)))
    lines.append("consts")
    want.append(f"{hx(ed.D)} {hx(ed.D2)}")
    # lazy sums as multiplication operands: the documented bound (one un-carried addition)
    big = [P - 1, 2**255 - 20, 2**256 - 1, 2**256 - 38, (1 << 255) - 1] + [rng.getrandbits(256) for _ in range(40)]
    for _ in range(200):
        a, b, c, d = (rng.choice(big) for _ in range(4))
        lines.append(f"felazy {hx(a)} {hx(b)} {hx(c)} {hx(d)}")
        want.append(f"{hx((a + b) * (c + d) % P)} {hx((a + b) ** 2 % P)} {hx((a + b - c - d) % P)}")
    assert harness(lines) == want


def test_host_51_bit_field(harness):
    """fe51_host.h - the prover's per-round inversion on the host (prover.hip p4_affine_pair)"""
    rng = random.Random(51)
    vals = EDGE + [rng.getrandbits(256) for _ in range(100)]
    lines, want = [], []
    for a in vals:
        b = rng.choice(vals)
        lines.append(f"h51mul {hx(a)} {hx(b)}")
        want.append(hx(a * b % P))
        if a % P:
            lines.append(f"h51inv {hx(a)}")
            want.append(hx(pow(a, P - 2, P)))
    assert harness(lines) == want


def test_host_commitment_fold(harness):
    """fe51_host.h curve code behind vmpc_ed25519_fold_commitment_host: A + c Q + c^2 B (compressed_pivot.py:66)
    against the oracle's point arithmetic, identity operands and the doubling case of the unified addition included"""
    rng = random.Random(66)

    def rnd_pt():
        return ed.pt_affine(ed.pt_repeat(ed.BASE, rng.randrange(1, ELL)))
    ident = (0, 1)
    cases = [(rnd_pt(), rnd_pt(), rnd_pt(), rng.randrange(ELL)) for _ in range(12)]
    cases += [(ident, rnd_pt(), rnd_pt(), rng.randrange(ELL)), (rnd_pt(), ident, ident, 5), (rnd_pt(), rnd_pt(), rnd_pt(), 0),
              (rnd_pt(), rnd_pt(), rnd_pt(), 1), (rnd_pt(), rnd_pt(), rnd_pt(), ELL - 1)]
    p0 = rnd_pt()
    cases.append((p0, p0, p0, 2))
    lines, want = [], []
    for A, Q, Bp, c in cases:
        lines.append("h51lin3 " + " ".join(hx(v) for v in (A[0], A[1], Q[0], Q[1], Bp[0], Bp[1], c)))
        ext = lambda a: (a[0], a[1], 1)
        R = ed.pt_add(ed.pt_add(ext(A), ed.pt_repeat(ext(Q), c)), ed.pt_repeat(ext(Bp), c * c))
        x, y = ed.pt_affine(R)
        want.append(f"{hx(x)} {hx(y)}")
    assert harness(lines) == want


def _limbs_value(limbs):
    return sum(v << ((51 * i + 1) // 2) for i, v in enumerate(limbs))


def test_field_mul_operand_contract(harness):
    """fe_mul(f, g) with every even limb of f just below 2^28 and of g just below 2^27.2 (product of
    maxima 2^55.2), fe_sqr with even limbs just below 2^27.6; odd limbs one bit less: the
    documented lazy-operand bounds of fe25519.h."""
    rng = random.Random(11)
    fmax, gmax, smax = (1 << 28) - 1, int(2 ** 27.2), int(2 ** 27.6)

    def vec(hi, full):
        return [(hi >> (i & 1)) if full or rng.random() < 0.5 else rng.randrange((hi >> (i & 1)) + 1)
                for i in range(10)]

    wide = (int(2 ** 28.4), (1 << 26) + (1 << 15))      # B - C - D against a reduced partner
    cases = [(vec(fmax, True), vec(gmax, True)), (vec(smax, True), vec(smax, True)),
             (vec(wide[0], True), vec(wide[1], True))]
    for _ in range(300):
        hi_f, hi_g = rng.choice([(fmax, gmax), (smax, smax), wide])   # asymmetric contract: 19*g < 2^32
        cases.append((vec(hi_f, False), vec(hi_g, False)))
    lines = ["rawmul " + " ".join(map(str, f)) + " " + " ".join(map(str, g)) for f, g in cases]
    for (f, g), o in zip(cases, harness(lines)):
        got_mul, got_sqr = o.split()
        assert got_mul == hx(_limbs_value(f) * _limbs_value(g) % P)
        if max(g[0::2]) <= smax:
            assert got_sqr == hx(_limbs_value(g) ** 2 % P)


def test_scalar_field_ops(harness):
    rng = random.Random(2)
    vals = [0, 1, 2, ELL - 1, ELL - 2, ELL // 2, ELL // 2 + 1, 2**252, 2**252 - 1] + \
           [rng.randrange(ELL) for _ in range(100)]
    lines, want = [], []
    for _ in range(400):
        a, b = rng.choice(vals), rng.choice(vals)
        lines += [f"fradd {hx(a)} {hx(b)}", f"frsub {hx(a)} {hx(b)}", f"frmul {hx(a)} {hx(b)}"]
        want += [hx((a + b) % ELL), hx((a - b) % ELL), hx(a * b % ELL)]
    for x in [0, 1, ELL, ELL - 1, 2**512 - 1, 2**511, ELL * ELL, (ELL - 1) ** 2, 2**256 - 1, 2**256] + \
            [rng.getrandbits(512) for _ in range(100)]:
        lines.append(f"frred {hx(x)}")
        want.append(hx(x % ELL))
    for a in vals:
        for sg in (0, 1):
            lines.append(f"frrepr {hx(a)} {sg}")
            s = str(a - ELL) if (sg and a > ELL // 2) else str(a)
            want.append(f"{s} {len(s)}")
    assert harness(lines) == want


def test_decimal(harness):
    rng = random.Random(3)
    vals = [0, 1, 9, 10, 10**9 - 1, 10**9, 10**18, 10**77, 10**77 - 1, 2**256 - 1, P - 1] + \
           [rng.getrandbits(rng.randrange(1, 257)) for _ in range(200)]
    out = harness([f"dec {hx(v)}" for v in vals])
    assert out == [f"{v} {len(str(v))}" for v in vals]


def rand_point(rng):
    return ed.pt_repeat(ed.BASE, rng.randrange(1, ELL))


def test_projective_replay(harness):
    """add-2008-bbjlp / dbl-2008-bbjlp / right-to-left repeat give the oracle's exact
    (X, Y, Z) representatives, not merely the same group element."""
    rng = random.Random(4)
    lines, want = [], []
    pts = [ed.IDENTITY, ed.BASE] + [rand_point(rng) for _ in range(6)]
    pts += [ed.pt_add(pts[2], pts[3]), ed.pt_dbl(pts[4])]          # Z != 1
    f3 = lambda t: " ".join(hx(c) for c in t)
    half = (ed.P - 1) // 2                  # the sign boundary of a coordinate: (p - 1) / 2 prints positive
    for fake in ((half, half + 1, ed.P - 1), (half - 1, 0, 1), (2**254 - 1, 2**254, 2**254 - 9)):
        for cmd, fmt in (("prepr", ("[]", False)), ("preprs", ("[]", True)), ("preprp", ("()", True))):
            ed.set_format(*fmt)
            try:
                lines.append(f"{cmd} {f3(fake)}")
                want.append(ed.pt_repr(fake) + "|" + str(len(ed.pt_repr(fake))))
            finally:
                ed.set_format()
    for p in pts:
        for q in pts[:5]:
            lines.append(f"padd {f3(p)} {f3(q)}")
            want.append(f3(ed.pt_add(p, q)))
        lines.append(f"pdbl {f3(p)}")
        want.append(f3(ed.pt_dbl(p)))
        lines.append(f"prepr {f3(p)}")
        want.append(ed.pt_repr(p) + "|" + str(len(ed.pt_repr(p))))
        # the other recalled formats of a point: signed coordinates, round brackets (runtime switches of fmt.h)
        for cmd, fmt in (("preprs", ("[]", True)), ("preprp", ("()", True))):
            ed.set_format(*fmt)
            try:
                lines.append(f"{cmd} {f3(p)}")
                want.append(ed.pt_repr(p) + "|" + str(len(ed.pt_repr(p))))
            finally:
                ed.set_format()
        for n in [0, 1, 2, 3, 2**32 - 1, 2**32, 2**33 + 1, ELL - 1, ELL, 2**253 - 1,
                  rng.randrange(ELL), rng.randrange(ELL) ** 2 % 2**256, rng.getrandbits(64)]:
            lines.append(f"prepeat {f3(p)} {hx(n)}")
            want.append(f3(ed.pt_repeat(p, n)))
    assert harness(lines) == want


def test_extended_formulas(harness):
    rng = random.Random(5)
    lines, want = [], []
    f2 = lambda t: " ".join(hx(c) for c in ed.pt_affine(t))
    pts = [rand_point(rng) for _ in range(8)]
    for p in pts:
        lines.append(f"edbl {f2(p)}")
        want.append(f2(ed.pt_repeat(p, 4)) + " 1")
        for q in pts[:4] + [p, ed.pt_repeat(p, 3), ed.pt_neg(ed.pt_repeat(p, 3)), ed.IDENTITY]:
            p3 = ed.pt_repeat(p, 3)
            lines.append(f"eadd {f2(p)} {f2(q)}")
            want.append(f2(ed.pt_add(p3, q)) + " 1")
            lines.append(f"emadd {f2(p)} {f2(q)}")
            want.append(f2(ed.pt_add(p3, q)) + " 1")
            lines.append(f"emaddneg {f2(p)} {f2(q)}")
            want.append(f2(ed.pt_add(p3, ed.pt_neg(q))) + " 1")
    assert harness(lines) == want


# ---- BN-256 ---------------------------------------------------------------------
from oracle import bn256_ref as bn


def test_bn256_field(harness):
    rng = random.Random(6)
    vals = [0, 1, 2, bn.P - 1, bn.P - 2, 2**255, 2**128 - 1] + [rng.randrange(bn.P) for _ in range(60)]
    lines, want = [], []
    for _ in range(300):
        a, b = rng.choice(vals), rng.choice(vals)
        lines += [f"bnmul {hx(a)} {hx(b)}", f"bnadd {hx(a)} {hx(b)}", f"bnsub {hx(a)} {hx(b)}"]
        want += [hx(a * b % bn.P), hx((a + b) % bn.P), hx((a - b) % bn.P)]
    for a in vals[1:20]:
        lines.append(f"bninv {hx(a)}")
        want.append(hx(pow(a, bn.P - 2, bn.P)))
    for _ in range(60):
        a = (rng.randrange(bn.P), rng.randrange(bn.P))
        b = (rng.randrange(bn.P), rng.randrange(bn.P))
        lines += [f"bn2mul {hx(a[0])} {hx(a[1])} {hx(b[0])} {hx(b[1])}", f"bn2sqr {hx(a[0])} {hx(a[1])}",
                  f"bn2inv {hx(a[0])} {hx(a[1])}"]
        want += [" ".join(hx(v) for v in bn.Fp2.mul(a, b)), " ".join(hx(v) for v in bn.Fp2.mul(a, a)),
                 " ".join(hx(v) for v in bn.Fp2.inv(a))]
    assert harness(lines) == want


def _flat(pt):
    if pt is None:
        return "inf"
    out = []
    for c in pt:
        out += list(c) if isinstance(c, tuple) else [c]
    return " ".join(hx(v) for v in out)


def test_bn256_latency_field(harness):
    """fp29.h (9 x 29-bit limbs, Montgomery radix 2^261: the field of the reduction / recombination kernels) against
    big ints: products, sums, differences with every value kept in [0, 2p), the zero test on unreduced values (p itself
    is a representation of zero), inversion, and the conversion from / to sw256.h's memory format"""
    rng = random.Random(29)
    vals = [0, 1, 2, bn.P - 1, bn.P - 2, 2**255, 2**128 - 1, (bn.P + 1) // 2] + [rng.randrange(bn.P) for _ in range(80)]
    lines, want = [], []
    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        lines += [f"l29mul {hx(a)} {hx(b)}", f"l29add {hx(a)} {hx(b)}", f"l29sub {hx(a)} {hx(b)}", f"l29sub {hx(a)} {hx(a)}"]
        want += [hx(a * b % bn.P), hx((a + b) % bn.P), hx((a - b) % bn.P), hx(0)]
        t = (((a + b) * (a - b) - a * a + b * b) * 2 * a) % bn.P          # == 0 identically
        lines.append(f"l29chain {hx(a)} {hx(b)}")
        want.append(f"{hx(t)} 1 {1 if a == b else 0}")
        lines.append(f"l29raw {hx(a)}")
        want.append(f"{hx(a)} 1")
        lines.append(f"l29sqr {hx(a)}")
        want.append(f"{hx(a * a % bn.P)} {hx(4 * a * a % bn.P)}")
    for a in vals[1:24]:
        lines.append(f"l29inv {hx(a)}")
        want.append(hx(pow(a, bn.P - 2, bn.P)))
    for _ in range(12):
        a = (rng.randrange(bn.P), rng.randrange(bn.P))
        b = (rng.randrange(bn.P), rng.randrange(bn.P))
        lines.append(f"l29x2mul {hx(a[0])} {hx(a[1])} {hx(b[0])} {hx(b[1])}")
        want.append(" ".join(hx(v) for v in bn.Fp2.mul(a, b) + bn.Fp2.mul(a, a) + bn.Fp2.inv(a)))
    assert harness(lines) == want


@pytest.mark.parametrize("which", ["g1", "g2", "h1", "h2"])
def test_bn256_curve_ops(harness, which):
    E, G = (bn.E1, bn.G1) if which in ("g1", "h1") else (bn.E2, bn.G2)       # h1 / h2: the same curves over fp29.h
    rng = random.Random(7)
    pts = [E.mul(rng.randrange(1, bn.N), G) for _ in range(5)]
    lines, want = [], []
    for p in pts:
        lines.append(f"{which}dbl {_flat(p)}")
        want.append(_flat(E.add(p, p)))
        for q in pts[:3] + [p, E.neg(p), E.add(p, p), E.neg(E.add(p, p))]:
            lines.append(f"{which}add {_flat(p)} {_flat(q)}")
            want.append(_flat(E.add(E.add(p, p), E.add(q, q))))
            lines.append(f"{which}madd {_flat(p)} {_flat(q)}")
            want.append(_flat(E.add(E.add(p, p), q)))
        for k in [0, 1, 2, bn.N - 1, bn.N, rng.randrange(bn.N), 2**256 - 1]:
            lines.append(f"{which}mul {_flat(p)} {hx(k)}")
            want.append(_flat(E.mul(k, p)))
    assert harness(lines) == want
