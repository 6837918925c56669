// BN-256 base field of the device kernels: 9 unsaturated limbs of 29 bits, Montgomery radix 2^261 (round 4).
//
// sw256.h's fp (8 saturated limbs, finely integrated product scan: 128 multiply-adds into ONE three-word accumulator -
// a serial chain) was the field of rounds 2-3.  Measured on gfx950 (scripts/bn_field_bench.hip): 742 ns per DEPENDENT
// product on one wave and 111 G products/s chip-wide.  Here a product is 81 + 81 plain 64-bit multiply-adds into 18
// INDEPENDENT column accumulators (29 + 29 bits x 18 terms fit 64 bits, as in fe25519.h) plus nine short reduction
// steps: 478 ns per dependent product and 150 G products/s - the latency chains of the bucket reduction and the
// recombination (half of a prepared-key sum; verifiable_mpc/trinocchio/pynocchio.py:228-246 is what those sums
// replace) AND the throughput-bound bucket pass gain.  With R = 2^261 >= 57 p no final subtraction is needed: operands
// < 2p give a result < 1.07 p.  (Round 3 had costed unsaturated limbs at "250 instructions again" and not written
// them: the instruction count is about the same - what differs is that the 162 multiply-adds have no carries between
// them.)
//
// Values: limbs normalised (< 2^29, the top one holds the rest), value in [0, 2p).  Workspace / table format: the
// canonical residue of x 2^261 as eight 32-bit words (load_raw / store_raw: shifts only).  sw256.h stays as the
// host-testable reference statement of the curve formulas and supplies the curve template jac<F>.
// VMPC_HD: host-testable (tests/native/host_math_test.cpp).
#pragma once
#include "sw256.h"

#define FP29_LIMBS 9
#define FP29_MASK 0x1fffffffu

struct fp29 {
    uint32_t v[FP29_LIMBS];
};

#define FP29_P                                                                                                   \
    { 0x1e089667u, 0x02e56362u, 0x0d6d6786u, 0x1711a241u, 0x0dc21ee5u, 0x165c30c2u, 0x1fe6a9bfu, 0x1c695470u,       \
      0x008fb501u }
#define FP29_2P                                                                                                  \
    { 0x1c112cceu, 0x05cac6c5u, 0x1adacf0cu, 0x0e234482u, 0x1b843dcbu, 0x0cb86184u, 0x1fcd537fu, 0x18d2a8e1u,       \
      0x011f6a03u }
#define FP29_N0 0x1f17daa9u      // -p^-1 mod 2^29
#define FP29_ONE                                                                                                 \
    { 0x10168311u, 0x1aecdef8u, 0x02a3f324u, 0x1d12df6fu, 0x0fc71ed9u, 0x057924b5u, 0x05a43451u, 0x0c8c32d7u,       \
      0x0000b294u }                 // 2^261 mod p
#define FP29_RR                                                                                                  \
    { 0x1d8d65edu, 0x157f5abfu, 0x058bb993u, 0x0a97ab55u, 0x043b578du, 0x0a4d6606u, 0x085e0882u, 0x1df0d5b9u,       \
      0x000ec411u }                 // 2^522 mod p: x -> (x 2^261)

VMPC_HD fp29 fp29_zero() {
    fp29 r;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) r.v[i] = 0;
    return r;
}
VMPC_HD fp29 fp29_one() {
    fp29 r = {FP29_ONE};
    return r;
}
VMPC_HD fp29 fp29_select(const fp29 &a, const fp29 &b, bool pick_b) {
    fp29 r;
    const uint32_t m = 0u - (uint32_t)pick_b;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}

// t (signed limbs, any size that fits 32 bits with carries) -> normalised limbs; the value must be >= 0
VMPC_HD fp29 fp29_normalise(const int32_t t[FP29_LIMBS]) {
    fp29 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS - 1; i++) {
        const int32_t s = t[i] + c;
        r.v[i] = (uint32_t)s & FP29_MASK;
        c = s >> 29;                    // arithmetic shift: borrows travel as negative carries
    }
    r.v[FP29_LIMBS - 1] = (uint32_t)(t[FP29_LIMBS - 1] + c);
    return r;
}

// a (normalised, < 4p) -> a - 2p if a >= 2p
VMPC_HD fp29 fp29_cond_sub_2p(const fp29 &a) {
    const uint32_t P2[FP29_LIMBS] = FP29_2P;
    int32_t d[FP29_LIMBS];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS - 1; i++) {
        const int32_t s = (int32_t)a.v[i] - (int32_t)P2[i] + c;
        d[i] = s & (int32_t)FP29_MASK;
        c = s >> 29;
    }
    d[FP29_LIMBS - 1] = (int32_t)a.v[FP29_LIMBS - 1] - (int32_t)P2[FP29_LIMBS - 1] + c;
    const bool neg = d[FP29_LIMBS - 1] < 0;           // a < 2p: keep a
    fp29 r;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) r.v[i] = neg ? a.v[i] : (uint32_t)d[i];
    return r;
}

VMPC_HD fp29 fp29_add(const fp29 &a, const fp29 &b) {
    int32_t t[FP29_LIMBS];
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) t[i] = (int32_t)(a.v[i] + b.v[i]);
    return fp29_cond_sub_2p(fp29_normalise(t));
}

VMPC_HD fp29 fp29_sub(const fp29 &a, const fp29 &b) {       // a - b + 2p, in (0, 4p)
    const uint32_t P2[FP29_LIMBS] = FP29_2P;
    int32_t t[FP29_LIMBS];
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) t[i] = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)P2[i];
    return fp29_cond_sub_2p(fp29_normalise(t));
}

VMPC_HD fp29 fp29_neg(const fp29 &a) { return fp29_sub(fp29_zero(), a); }
VMPC_HD fp29 fp29_dbl(const fp29 &a) { return fp29_add(a, a); }

// Montgomery reduction of the 18 column sums t (each < 2^62.2 after the 81 reduction products are added): nine
// steps, then the carry pass over the upper half.  Result < 1.07 p for a product of operands < 2p.
VMPC_HD fp29 fp29_mont_reduce(uint64_t t[2 * FP29_LIMBS]) {
    const uint32_t Pl[FP29_LIMBS] = FP29_P;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) {
        const uint32_t m = ((uint32_t)t[i] * FP29_N0) & FP29_MASK;
#pragma unroll
        for (int j = 0; j < FP29_LIMBS; j++) t[i + j] += (uint64_t)m * Pl[j];
        t[i + 1] += t[i] >> 29;          // the low 29 bits of t[i] are zero now
    }
    fp29 r;
#pragma unroll
    for (int j = 0; j < FP29_LIMBS - 1; j++) {
        r.v[j] = (uint32_t)t[FP29_LIMBS + j] & FP29_MASK;
        t[FP29_LIMBS + j + 1] += t[FP29_LIMBS + j] >> 29;
    }
    r.v[FP29_LIMBS - 1] = (uint32_t)t[2 * FP29_LIMBS - 1];
    return r;
}

// Montgomery product a b 2^-261 mod p, operands < 2p, result < 1.07 p
VMPC_HD fp29 fp29_mul(const fp29 &a, const fp29 &b) {
    uint64_t t[2 * FP29_LIMBS];
#pragma unroll
    for (int k = 0; k < 2 * FP29_LIMBS; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) {
#pragma unroll
        for (int j = 0; j < FP29_LIMBS; j++) t[i + j] += (uint64_t)a.v[i] * b.v[j];
    }
    return fp29_mont_reduce(t);
}

// a^2 2^-261: 45 products (the off-diagonal ones against the doubled limbs: 2^30 x 2^29 x 9 terms still fit 64 bits)
VMPC_HD fp29 fp29_sqr(const fp29 &a) {
    uint32_t a2[FP29_LIMBS];
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) a2[i] = 2u * a.v[i];
    uint64_t t[2 * FP29_LIMBS];
#pragma unroll
    for (int k = 0; k < 2 * FP29_LIMBS; k++) t[k] = 0;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) {
        t[2 * i] += (uint64_t)a.v[i] * a.v[i];
#pragma unroll
        for (int j = i + 1; j < FP29_LIMBS; j++) t[i + j] += (uint64_t)a2[i] * a.v[j];
    }
    return fp29_mont_reduce(t);
}

// a in [0, 2p) -> the residue in [0, p)
VMPC_HD fp29 fp29_canon(const fp29 &a) {
    const uint32_t Pl[FP29_LIMBS] = FP29_P;
    int32_t d[FP29_LIMBS];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS - 1; i++) {
        const int32_t s = (int32_t)a.v[i] - (int32_t)Pl[i] + c;
        d[i] = s & (int32_t)FP29_MASK;
        c = s >> 29;
    }
    d[FP29_LIMBS - 1] = (int32_t)a.v[FP29_LIMBS - 1] - (int32_t)Pl[FP29_LIMBS - 1] + c;
    const bool neg = d[FP29_LIMBS - 1] < 0;
    fp29 r;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) r.v[i] = neg ? a.v[i] : (uint32_t)d[i];
    return r;
}

VMPC_HD bool fp29_is_zero(const fp29 &a) {       // a in [0, 2p): zero as a residue <=> a == 0 or a == p
    const uint32_t Pl[FP29_LIMBS] = FP29_P;
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < FP29_LIMBS; i++) {
        z |= a.v[i];
        e |= a.v[i] ^ Pl[i];
    }
    return z == 0 || e == 0;
}

// a^(p-2): left-to-right square-and-multiply over the 256 bits of p - 2
VMPC_HD fp29 fp29_inv(const fp29 &a) {
    const uint32_t Pw[8] = BN_P_LIMBS;
    fp29 r = fp29_one();
    for (int w = 7; w >= 0; w--) {
        const uint32_t word = w == 0 ? Pw[0] - 2u : Pw[w];       // p is odd and p[0] >= 2: no borrow
        for (int b = 31; b >= 0; b--) {
            r = fp29_sqr(r);
            if ((word >> b) & 1u) r = fp29_mul(r, a);
        }
    }
    return r;
}

// eight 32-bit words (an integer < 2^256) <-> nine 29-bit limbs
VMPC_HD fp29 fp29_from_words(const uint32_t w[8]) {
    fp29 r;
#pragma unroll
    for (int j = 0; j < FP29_LIMBS; j++) {
        const int bit = 29 * j, word = bit >> 5, off = bit & 31;
        uint32_t lo = w[word] >> off;
        if (off > 3 && word + 1 < 8) lo |= w[word + 1] << (32 - off);   // the limb straddles two words
        r.v[j] = j < FP29_LIMBS - 1 ? (lo & FP29_MASK) : lo;
    }
    return r;
}
VMPC_HD void fp29_to_words(const fp29 &a, uint32_t w[8]) {     // a normalised, < 2^256
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = 0;
#pragma unroll
    for (int j = 0; j < FP29_LIMBS; j++) {
        const int bit = 29 * j, word = bit >> 5, off = bit & 31;
        w[word] |= a.v[j] << off;
        if (off > 3 && word + 1 < 8) w[word + 1] |= a.v[j] >> (32 - off);
    }
}

// ---- F_p2 = F_p[i]/(i^2+1) over fp29 --------------------------------------------------------------------------------
struct fp29x2 {
    fp29 a, b;
};
VMPC_HD fp29x2 fp29x2_mul(const fp29x2 &x, const fp29x2 &y) {   // Karatsuba: 3 base products
    const fp29 t0 = fp29_mul(x.a, y.a), t1 = fp29_mul(x.b, y.b);
    const fp29 t2 = fp29_mul(fp29_add(x.a, x.b), fp29_add(y.a, y.b));
    fp29x2 r;
    r.a = fp29_sub(t0, t1);
    r.b = fp29_sub(fp29_sub(t2, t0), t1);
    return r;
}
VMPC_HD fp29x2 fp29x2_sqr(const fp29x2 &x) {                    // (a+b)(a-b) + 2ab i
    fp29x2 r;
    r.a = fp29_mul(fp29_add(x.a, x.b), fp29_sub(x.a, x.b));
    r.b = fp29_dbl(fp29_mul(x.a, x.b));
    return r;
}

// uniform field interfaces for the curve template (sw256.h jac<F>): same memory format as Fp1Ops / Fp2Ops
struct Fp29Ops {
    typedef fp29 elem;
    static constexpr int WORDS = 8;
    VMPC_HD static elem zero() { return fp29_zero(); }
    VMPC_HD static elem one() { return fp29_one(); }
    VMPC_HD static elem add(const elem &x, const elem &y) { return fp29_add(x, y); }
    VMPC_HD static elem sub(const elem &x, const elem &y) { return fp29_sub(x, y); }
    VMPC_HD static elem mul(const elem &x, const elem &y) { return fp29_mul(x, y); }
    VMPC_HD static elem sqr(const elem &x) { return fp29_sqr(x); }
    VMPC_HD static elem neg(const elem &x) { return fp29_neg(x); }
    VMPC_HD static elem dbl(const elem &x) { return fp29_dbl(x); }
    VMPC_HD static elem inv(const elem &x) { return fp29_inv(x); }
    VMPC_HD static bool is_zero(const elem &x) { return fp29_is_zero(x); }
    VMPC_HD static elem select(const elem &x, const elem &y, bool p) { return fp29_select(x, y, p); }
    VMPC_HD static elem load(const uint32_t *src) {            // canonical LE -> x 2^261
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = src[i];
        const fp29 rr = {FP29_RR};
        return fp29_mul(fp29_from_words(w), rr);
    }
    VMPC_HD static void store(uint32_t *dst, const elem &x) {   // x 2^261 -> canonical LE
        fp29 one = fp29_zero();
        one.v[0] = 1;
        uint32_t w[8];
        fp29_to_words(fp29_canon(fp29_mul(x, one)), w);
#pragma unroll
        for (int i = 0; i < 8; i++) dst[i] = w[i];
    }
    // workspace / table format: the canonical residue of x 2^261 in eight 32-bit words - shifts only, no product
    VMPC_HD static elem load_raw(const uint32_t *src) {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = src[i];
        return fp29_from_words(w);
    }
    VMPC_HD static void store_raw(uint32_t *dst, const elem &x) {
        uint32_t w[8];
        fp29_to_words(fp29_canon(x), w);
#pragma unroll
        for (int i = 0; i < 8; i++) dst[i] = w[i];
    }
    VMPC_HD static bool raw_canonical(const uint32_t *src) {   // a canonical residue (< p) as a plain integer
        fp r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = src[i];
        return fp_raw_is_canonical(r);
    }
};
struct Fp29x2Ops {
    typedef fp29x2 elem;
    static constexpr int WORDS = 16;
    VMPC_HD static elem zero() {
        elem r;
        r.a = fp29_zero();
        r.b = fp29_zero();
        return r;
    }
    VMPC_HD static elem one() {
        elem r;
        r.a = fp29_one();
        r.b = fp29_zero();
        return r;
    }
    VMPC_HD static elem add(const elem &x, const elem &y) {
        elem r;
        r.a = fp29_add(x.a, y.a);
        r.b = fp29_add(x.b, y.b);
        return r;
    }
    VMPC_HD static elem sub(const elem &x, const elem &y) {
        elem r;
        r.a = fp29_sub(x.a, y.a);
        r.b = fp29_sub(x.b, y.b);
        return r;
    }
    VMPC_HD static elem mul(const elem &x, const elem &y) { return fp29x2_mul(x, y); }
    VMPC_HD static elem sqr(const elem &x) { return fp29x2_sqr(x); }
    VMPC_HD static elem neg(const elem &x) {
        elem r;
        r.a = fp29_neg(x.a);
        r.b = fp29_neg(x.b);
        return r;
    }
    VMPC_HD static elem dbl(const elem &x) { return add(x, x); }
    VMPC_HD static elem inv(const elem &x) {
        const fp29 d = fp29_inv(fp29_add(fp29_sqr(x.a), fp29_sqr(x.b)));
        elem r;
        r.a = fp29_mul(x.a, d);
        r.b = fp29_neg(fp29_mul(x.b, d));
        return r;
    }
    VMPC_HD static bool is_zero(const elem &x) { return fp29_is_zero(x.a) && fp29_is_zero(x.b); }
    VMPC_HD static elem select(const elem &x, const elem &y, bool p) {
        elem r;
        r.a = fp29_select(x.a, y.a, p);
        r.b = fp29_select(x.b, y.b, p);
        return r;
    }
    VMPC_HD static elem load(const uint32_t *src) {
        elem r;
        r.a = Fp29Ops::load(src);
        r.b = Fp29Ops::load(src + 8);
        return r;
    }
    VMPC_HD static void store(uint32_t *dst, const elem &x) {
        Fp29Ops::store(dst, x.a);
        Fp29Ops::store(dst + 8, x.b);
    }
    VMPC_HD static elem load_raw(const uint32_t *src) {
        elem r;
        r.a = Fp29Ops::load_raw(src);
        r.b = Fp29Ops::load_raw(src + 8);
        return r;
    }
    VMPC_HD static void store_raw(uint32_t *dst, const elem &x) {
        Fp29Ops::store_raw(dst, x.a);
        Fp29Ops::store_raw(dst + 8, x.b);
    }
    VMPC_HD static bool raw_canonical(const uint32_t *src) {
        return Fp29Ops::raw_canonical(src) && Fp29Ops::raw_canonical(src + 8);
    }
};
