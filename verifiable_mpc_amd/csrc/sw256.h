// BN-256 field and curve arithmetic for gfx950 (BASELINE config 5):
//   fp    F_p, p = 36u^4+36u^3+24u^2+6u+1, u = 1868033^3 (verifiable_mpc/ac20/pairing.py:49-51),
//         8 x 32-bit limbs in MONTGOMERY form (R = 2^256); p has no special shape, so the
//         product is a finely-integrated product scan: 64 + 64 multiply-adds per multiplication
//   fp2   F_p[i]/(i^2+1)  (the twist's field; xi = i + 3)
//   jac<F> Jacobian points on y^2 = x^3 + b (a = 0) over F = fp (G1) or fp2 (G2)
// Replaces the MPyC EllipticCurve('BN256' / 'BN256_twist', 'jacobian') arithmetic behind
//   [int(c[i]) * evalkey[...]] + apply_to_list(point_add, ...)
//   verifiable_mpc/trinocchio/pynocchio.py:229-246.
// Only affine results are defined as output (proof elements feed pairings).
// VMPC_HD: host-testable (tests/native/host_math_test.cpp).
#pragma once
#include "fe25519.h"   // VMPC_HD, fe_mac96 (device)

struct fp {
    uint32_t v[8];
};

#define BN_P_LIMBS                                                                             \
    { 0x5e089667u, 0x185cac6cu, 0x20b5b59eu, 0xee5b88d1u, 0x6184dc21u, 0xaa6fecb8u, 0x4aa387f9u,  \
      0x8fb501e3u }
// -p^{-1} mod 2^32
#define BN_N0 0x7f17daa9u
// R mod p (Montgomery one), R^2 mod p
#define BN_R1                                                                                  \
    { 0xa1f76999u, 0xe7a35393u, 0xdf4a4a61u, 0x11a4772eu, 0x9e7b23deu, 0x55901347u, 0xb55c7806u,  \
      0x704afe1cu }
#define BN_R2                                                                                  \
    { 0x7e444f56u, 0x9c21c3ffu, 0xb2efb0c2u, 0x409ed151u, 0x80fb1651u, 0x0c6dc37bu, 0x2c2380b7u,  \
      0x7c36e0e6u }

VMPC_HD fp fp_zero() {
    fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
VMPC_HD fp fp_one() {
    fp r = {BN_R1};
    return r;
}
VMPC_HD bool fp_is_zero(const fp &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
VMPC_HD bool fp_eq(const fp &a, const fp &b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
VMPC_HD fp fp_select(const fp &a, const fp &b, bool pick_b) {
    fp r;
    uint32_t m = 0u - (uint32_t)pick_b;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}

// r = t - p if (carry:t) >= p else t     (t < 2p)
VMPC_HD fp fp_cond_sub_p(const uint32_t t[8], uint32_t carry) {
    const uint32_t Pl[8] = BN_P_LIMBS;
    uint32_t s[8];
    int64_t b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        b += (int64_t)t[i] - (int64_t)Pl[i];
        s[i] = (uint32_t)b;
        b >>= 32;
    }
    // keep t only if there was a borrow and no incoming carry
    bool keep = (b != 0) && (carry == 0);
    uint32_t m = 0u - (uint32_t)keep;
    fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (t[i] & m) | (s[i] & ~m);
    return r;
}

VMPC_HD fp fp_add(const fp &a, const fp &b) {
    uint32_t t[8];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    return fp_cond_sub_p(t, (uint32_t)c);
}

VMPC_HD fp fp_sub(const fp &a, const fp &b) {
    const uint32_t Pl[8] = BN_P_LIMBS;
    fp r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    uint32_t m = (uint32_t)c;   // all ones on borrow: add p back
    uint64_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d += (uint64_t)r.v[i] + (Pl[i] & m);
        r.v[i] = (uint32_t)d;
        d >>= 32;
    }
    return r;
}

VMPC_HD fp fp_neg(const fp &a) { return fp_sub(fp_zero(), a); }
VMPC_HD fp fp_dbl(const fp &a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod p, finely integrated product scanning
VMPC_HD fp fp_mul(const fp &a, const fp &b) {
    const uint32_t Pl[8] = BN_P_LIMBS;
    uint32_t m[8], t[8];
#ifdef VMPC_DEVICE_ASM
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) fe_mac96(acc, ovf, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) fe_mac96(acc, ovf, m[i], Pl[k - i]);
        m[k] = (uint32_t)acc * BN_N0;
        fe_mac96(acc, ovf, m[k], Pl[0]);
        acc = (acc >> 32) | ((uint64_t)ovf << 32);
        ovf = 0;
    }
#pragma unroll
    for (int k = 8; k < 15; k++) {
#pragma unroll
        for (int i = k - 7; i < 8; i++) fe_mac96(acc, ovf, a.v[i], b.v[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; i++) fe_mac96(acc, ovf, m[i], Pl[k - i]);
        t[k - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ovf << 32);
        ovf = 0;
    }
    t[7] = (uint32_t)acc;
    return fp_cond_sub_p(t, (uint32_t)(acc >> 32));
#else
    // portable form of the same scan with a 3-word accumulator
    uint64_t lo = 0;     // low 64 bits
    uint32_t hi = 0;     // bits 64..95
#define BN_MAC(x, y)                                         \
    do {                                                     \
        uint64_t _p = (uint64_t)(x) * (y);                   \
        uint64_t _s = lo + _p;                               \
        hi += (uint32_t)(_s < lo);                           \
        lo = _s;                                             \
    } while (0)
    for (int k = 0; k < 8; k++) {
        for (int i = 0; i <= k; i++) BN_MAC(a.v[i], b.v[k - i]);
        for (int i = 0; i < k; i++) BN_MAC(m[i], Pl[k - i]);
        m[k] = (uint32_t)lo * BN_N0;
        BN_MAC(m[k], Pl[0]);
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    for (int k = 8; k < 15; k++) {
        for (int i = k - 7; i < 8; i++) BN_MAC(a.v[i], b.v[k - i]);
        for (int i = k - 7; i < 8; i++) BN_MAC(m[i], Pl[k - i]);
        t[k - 8] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#undef BN_MAC
    t[7] = (uint32_t)lo;
    return fp_cond_sub_p(t, (uint32_t)(lo >> 32));
#endif
}

VMPC_HD fp fp_sqr(const fp &a) { return fp_mul(a, a); }

VMPC_HD fp fp_to_mont(const fp &a) {
    fp r2 = {BN_R2};
    return fp_mul(a, r2);
}
VMPC_HD fp fp_from_mont(const fp &a) {
    fp one = fp_zero();
    one.v[0] = 1;
    return fp_mul(a, one);
}
VMPC_HD bool fp_raw_is_canonical(const fp &a) {   // a < p as a plain integer
    const uint32_t Pl[8] = BN_P_LIMBS;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] < Pl[i]) return true;
        if (a.v[i] > Pl[i]) return false;
    }
    return false;
}

// a^(p-2): plain left-to-right square-and-multiply over the 256 bits of p-2
VMPC_HD fp fp_inv(const fp &a) {
    const uint32_t Pl[8] = BN_P_LIMBS;
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = Pl[i];
    e[0] -= 2;   // p is odd and p[0] >= 2: no borrow
    fp r = fp_one();
#pragma unroll
    for (int w = 7; w >= 0; w--) {
        uint32_t word = e[w];
        for (int b = 31; b >= 0; b--) {
            r = fp_sqr(r);
            if ((word >> b) & 1u) r = fp_mul(r, a);
        }
    }
    return r;
}

// ---- F_p2 = F_p[i]/(i^2+1) ------------------------------------------------------------------------
struct fp2 {
    fp a, b;   // a + b i
};
VMPC_HD fp2 fp2_zero() {
    fp2 r;
    r.a = fp_zero();
    r.b = fp_zero();
    return r;
}
VMPC_HD fp2 fp2_one() {
    fp2 r;
    r.a = fp_one();
    r.b = fp_zero();
    return r;
}
VMPC_HD bool fp2_is_zero(const fp2 &x) { return fp_is_zero(x.a) && fp_is_zero(x.b); }
VMPC_HD bool fp2_eq(const fp2 &x, const fp2 &y) { return fp_eq(x.a, y.a) && fp_eq(x.b, y.b); }
VMPC_HD fp2 fp2_add(const fp2 &x, const fp2 &y) {
    fp2 r;
    r.a = fp_add(x.a, y.a);
    r.b = fp_add(x.b, y.b);
    return r;
}
VMPC_HD fp2 fp2_sub(const fp2 &x, const fp2 &y) {
    fp2 r;
    r.a = fp_sub(x.a, y.a);
    r.b = fp_sub(x.b, y.b);
    return r;
}
VMPC_HD fp2 fp2_neg(const fp2 &x) {
    fp2 r;
    r.a = fp_neg(x.a);
    r.b = fp_neg(x.b);
    return r;
}
VMPC_HD fp2 fp2_dbl(const fp2 &x) { return fp2_add(x, x); }
VMPC_HD fp2 fp2_mul(const fp2 &x, const fp2 &y) {   // Karatsuba: 3 base multiplications
    fp t0 = fp_mul(x.a, y.a), t1 = fp_mul(x.b, y.b);
    fp t2 = fp_mul(fp_add(x.a, x.b), fp_add(y.a, y.b));
    fp2 r;
    r.a = fp_sub(t0, t1);
    r.b = fp_sub(fp_sub(t2, t0), t1);
    return r;
}
VMPC_HD fp2 fp2_sqr(const fp2 &x) {   // (a+b)(a-b) + 2ab i
    fp2 r;
    r.a = fp_mul(fp_add(x.a, x.b), fp_sub(x.a, x.b));
    r.b = fp_dbl(fp_mul(x.a, x.b));
    return r;
}
VMPC_HD fp2 fp2_inv(const fp2 &x) {
    fp d = fp_inv(fp_add(fp_sqr(x.a), fp_sqr(x.b)));
    fp2 r;
    r.a = fp_mul(x.a, d);
    r.b = fp_neg(fp_mul(x.b, d));
    return r;
}
VMPC_HD fp2 fp2_select(const fp2 &x, const fp2 &y, bool pick_y) {
    fp2 r;
    r.a = fp_select(x.a, y.a, pick_y);
    r.b = fp_select(x.b, y.b, pick_y);
    return r;
}

// uniform field interface for the curve template
struct Fp1Ops {
    typedef fp elem;
    static constexpr int WORDS = 8;
    VMPC_HD static elem zero() { return fp_zero(); }
    VMPC_HD static elem one() { return fp_one(); }
    VMPC_HD static elem add(const elem &x, const elem &y) { return fp_add(x, y); }
    VMPC_HD static elem sub(const elem &x, const elem &y) { return fp_sub(x, y); }
    VMPC_HD static elem mul(const elem &x, const elem &y) { return fp_mul(x, y); }
    VMPC_HD static elem sqr(const elem &x) { return fp_sqr(x); }
    VMPC_HD static elem neg(const elem &x) { return fp_neg(x); }
    VMPC_HD static elem dbl(const elem &x) { return fp_dbl(x); }
    VMPC_HD static elem inv(const elem &x) { return fp_inv(x); }
    VMPC_HD static bool is_zero(const elem &x) { return fp_is_zero(x); }
    VMPC_HD static elem select(const elem &x, const elem &y, bool p) { return fp_select(x, y, p); }
    VMPC_HD static elem load(const uint32_t *src) {   // canonical LE -> Montgomery
        fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = src[i];
        return fp_to_mont(r);
    }
    VMPC_HD static void store(uint32_t *dst, const elem &x) {   // Montgomery -> canonical LE
        fp r = fp_from_mont(x);
#pragma unroll
        for (int i = 0; i < 8; i++) dst[i] = r.v[i];
    }
    VMPC_HD static elem load_raw(const uint32_t *src) {
        fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = src[i];
        return r;
    }
    VMPC_HD static void store_raw(uint32_t *dst, const elem &x) {
#pragma unroll
        for (int i = 0; i < 8; i++) dst[i] = x.v[i];
    }
    VMPC_HD static bool raw_canonical(const uint32_t *src) {
        fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = src[i];
        return fp_raw_is_canonical(r);
    }
};
struct Fp2Ops {
    typedef fp2 elem;
    static constexpr int WORDS = 16;
    VMPC_HD static elem zero() { return fp2_zero(); }
    VMPC_HD static elem one() { return fp2_one(); }
    VMPC_HD static elem add(const elem &x, const elem &y) { return fp2_add(x, y); }
    VMPC_HD static elem sub(const elem &x, const elem &y) { return fp2_sub(x, y); }
    VMPC_HD static elem mul(const elem &x, const elem &y) { return fp2_mul(x, y); }
    VMPC_HD static elem sqr(const elem &x) { return fp2_sqr(x); }
    VMPC_HD static elem neg(const elem &x) { return fp2_neg(x); }
    VMPC_HD static elem dbl(const elem &x) { return fp2_dbl(x); }
    VMPC_HD static elem inv(const elem &x) { return fp2_inv(x); }
    VMPC_HD static bool is_zero(const elem &x) { return fp2_is_zero(x); }
    VMPC_HD static elem select(const elem &x, const elem &y, bool p) { return fp2_select(x, y, p); }
    VMPC_HD static elem load(const uint32_t *src) {
        fp2 r;
        r.a = Fp1Ops::load(src);
        r.b = Fp1Ops::load(src + 8);
        return r;
    }
    VMPC_HD static void store(uint32_t *dst, const elem &x) {
        Fp1Ops::store(dst, x.a);
        Fp1Ops::store(dst + 8, x.b);
    }
    VMPC_HD static elem load_raw(const uint32_t *src) {
        fp2 r;
        r.a = Fp1Ops::load_raw(src);
        r.b = Fp1Ops::load_raw(src + 8);
        return r;
    }
    VMPC_HD static void store_raw(uint32_t *dst, const elem &x) {
        Fp1Ops::store_raw(dst, x.a);
        Fp1Ops::store_raw(dst + 8, x.b);
    }
    VMPC_HD static bool raw_canonical(const uint32_t *src) {
        return Fp1Ops::raw_canonical(src) && Fp1Ops::raw_canonical(src + 8);
    }
};

// ---- Jacobian points on y^2 = x^3 + b (a = 0); infinity is Z = 0 ---------------------------------
template <class F>
struct jac {
    typename F::elem X, Y, Z;
};
template <class F>
struct aff {                    // affine point in Montgomery form; inf marks the point at infinity
    typename F::elem x, y;
    bool inf;
};

template <class F>
VMPC_HD jac<F> jac_identity() {
    jac<F> r;
    r.X = F::one();
    r.Y = F::one();
    r.Z = F::zero();
    return r;
}

template <class F>
VMPC_HD jac<F> jac_select(const jac<F> &a, const jac<F> &b, bool pick_b) {
    jac<F> r;
    r.X = F::select(a.X, b.X, pick_b);
    r.Y = F::select(a.Y, b.Y, pick_b);
    r.Z = F::select(a.Z, b.Z, pick_b);
    return r;
}

// dbl-2009-l (a = 0): 2M + 5S.  Doubling infinity (Z = 0) or a 2-torsion point gives Z3 = 0.
template <class F>
VMPC_HD jac<F> jac_dbl(const jac<F> &p) {
    typedef typename F::elem E;
    E A = F::sqr(p.X);
    E B = F::sqr(p.Y);
    E C = F::sqr(B);
    E t = F::sqr(F::add(p.X, B));
    E D = F::dbl(F::sub(F::sub(t, A), C));
    E Ee = F::add(F::dbl(A), A);
    E Fq = F::sqr(Ee);
    jac<F> r;
    r.X = F::sub(Fq, F::dbl(D));
    E C8 = F::dbl(F::dbl(F::dbl(C)));
    r.Y = F::sub(F::mul(Ee, F::sub(D, r.X)), C8);
    r.Z = F::dbl(F::mul(p.Y, p.Z));
    return r;
}

// mixed addition p + q, q affine (madd-2007-bl): 7M + 4S, all special cases handled
template <class F>
VMPC_HD jac<F> jac_madd(const jac<F> &p, const aff<F> &q) {
    typedef typename F::elem E;
    if (q.inf) return p;
    jac<F> qj;
    qj.X = q.x;
    qj.Y = q.y;
    qj.Z = F::one();
    if (F::is_zero(p.Z)) return qj;
    E Z1Z1 = F::sqr(p.Z);
    E U2 = F::mul(q.x, Z1Z1);
    E S2 = F::mul(F::mul(q.y, p.Z), Z1Z1);
    E H = F::sub(U2, p.X);
    E rr = F::dbl(F::sub(S2, p.Y));
    if (F::is_zero(H)) {
        if (F::is_zero(rr)) return jac_dbl<F>(qj);
        return jac_identity<F>();
    }
    E HH = F::sqr(H);
    E I = F::dbl(F::dbl(HH));
    E J = F::mul(H, I);
    E V = F::mul(p.X, I);
    jac<F> r;
    r.X = F::sub(F::sub(F::sqr(rr), J), F::dbl(V));
    r.Y = F::sub(F::mul(rr, F::sub(V, r.X)), F::dbl(F::mul(p.Y, J)));
    r.Z = F::sub(F::sub(F::sqr(F::add(p.Z, H)), Z1Z1), HH);
    return r;
}

// general addition (add-2007-bl): 11M + 5S, all special cases handled
template <class F>
VMPC_HD jac<F> jac_add(const jac<F> &p, const jac<F> &q) {
    typedef typename F::elem E;
    if (F::is_zero(p.Z)) return q;
    if (F::is_zero(q.Z)) return p;
    E Z1Z1 = F::sqr(p.Z), Z2Z2 = F::sqr(q.Z);
    E U1 = F::mul(p.X, Z2Z2), U2 = F::mul(q.X, Z1Z1);
    E S1 = F::mul(F::mul(p.Y, q.Z), Z2Z2), S2 = F::mul(F::mul(q.Y, p.Z), Z1Z1);
    E H = F::sub(U2, U1);
    E rr = F::dbl(F::sub(S2, S1));
    if (F::is_zero(H)) {
        if (F::is_zero(rr)) return jac_dbl<F>(p);
        return jac_identity<F>();
    }
    E I = F::sqr(F::dbl(H));
    E J = F::mul(H, I);
    E V = F::mul(U1, I);
    jac<F> r;
    r.X = F::sub(F::sub(F::sqr(rr), J), F::dbl(V));
    r.Y = F::sub(F::mul(rr, F::sub(V, r.X)), F::dbl(F::mul(S1, J)));
    r.Z = F::mul(F::sub(F::sub(F::sqr(F::add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
    return r;
}

template <class F>
VMPC_HD aff<F> jac_to_affine(const jac<F> &p) {
    aff<F> r;
    r.inf = F::is_zero(p.Z);
    if (r.inf) {
        r.x = F::zero();
        r.y = F::zero();
        return r;
    }
    typename F::elem zi = F::inv(p.Z);
    typename F::elem zi2 = F::sqr(zi);
    r.x = F::mul(p.X, zi2);
    r.y = F::mul(F::mul(p.Y, zi2), zi);
    return r;
}

// affine point <-> memory (canonical little-endian; all-zero bytes = infinity)
template <class F>
VMPC_HD aff<F> aff_load(const uint32_t *src) {
    aff<F> r;
    uint32_t o = 0;
    for (int i = 0; i < 2 * F::WORDS; i++) o |= src[i];
    r.inf = (o == 0);
    r.x = F::load(src);
    r.y = F::load(src + F::WORDS);
    return r;
}
template <class F>
VMPC_HD void aff_store(uint32_t *dst, const aff<F> &a) {
    if (a.inf) {
        for (int i = 0; i < 2 * F::WORDS; i++) dst[i] = 0;
        return;
    }
    F::store(dst, a.x);
    F::store(dst + F::WORDS, a.y);
}
