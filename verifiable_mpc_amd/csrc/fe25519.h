// GF(2^255-19) arithmetic for gfx950: 10 unsaturated limbs, radix 2^25.5.
//
// Replaces the MPyC prime-field element arithmetic that every curve operation of the
// reference's hot path bottoms out in (verifiable_mpc/ac20/pivot.py:143-144 ->
// mpyc.fingroups / mpyc.finfields; SURVEY.md section 8a "EllipticCurve element type").
//
// Why unsaturated limbs: on gfx950 a 32x32->64 multiply-add (v_mad_u64_u32) is the only wide
// multiplier, and CARRIES are what is expensive - a saturated 8 x 32-bit product costs, per
// partial product, the mad plus a v_addc through the carry-out plus a hazard s_nop plus register
// pair shuffling (measured: ~385 instructions per multiplication inside the MSM bucket kernel).
// With limbs of 26/25 bits (value = sum v[i] * 2^ceil(25.5 i)) ten partial products fit a
// 64-bit accumulator without overflow, the wrap-around of 2^255 = 19 is folded into the
// operands (19 * g_j), and a product is 100 bare multiply-adds plus one short carry pass;
// a squaring is 55.  Additions are limb-wise.
//
// Limb bounds.  "reduced" means v[even] <= 2^26 + 2^15, v[odd] <= 2^25 + 2^15: what fe_mul,
// fe_sqr, fe_add, fe_sub, fe_neg, fe_unpack return.  The *_lazy forms skip the carry pass:
//   fe_add_lazy(a, b)   limbs a + b
//   fe_sub_lazy(a, b)   limbs a + 2p - b      (b reduced)
// Operand contract of fe_mul(f, g), stated for EVEN limbs (odd limbs: one bit less, which every
// value built from reduced values and the 2p bias satisfies): f < 2^31, g < 2^27.7 with
// max(f) * max(g) <= 2^55.2 (then the ten 64-bit column sums, each term carrying at most the
// factors 19 * 2, stay below 2^64, and 19 * g_j fits 32 bits).  Reduced values, one lazy sum or
// difference of reduced values (< 2^27.6), and `lazy +- reduced` against a reduced partner all
// satisfy it; the call sites in ge25519.h state their bounds.  fe_sqr(f): limbs < 2^27.6.
// tests/native/host_math_test.cpp drives both at these bounds (command `rawmul`).
//
// Memory format (`fe8`): 32 bytes little-endian = 8 LE uint32 words, canonical residue < p on
// every public buffer (include/vmpc.h).  fe_unpack / fe_pack convert; internal workspaces keep
// the 10 limbs (40 bytes per element).
//
// All functions are VMPC_HD so the same source is unit-tested on the host
// (tests/native/host_math_test.cpp) against the Python oracle.  No MFMA use (255-bit modular
// integers).
#pragma once
#include <stdint.h>

#ifndef VMPC_HD
#if defined(__HIPCC__)
#define VMPC_HD __host__ __device__ __forceinline__
#else
#define VMPC_HD inline
#endif
#endif

#define FE_LIMBS 10

struct fe {
    uint32_t v[FE_LIMBS];
};

#define FE_MASK26 0x3ffffffu
#define FE_MASK25 0x1ffffffu

VMPC_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.v[i] = 0;
    return r;
}

VMPC_HD fe fe_one() {
    fe r = fe_zero();
    r.v[0] = 1;
    return r;
}

VMPC_HD fe fe_from_u32(uint32_t x) {
    fe r = fe_zero();
    r.v[0] = x & FE_MASK26;
    r.v[1] = x >> 26;
    return r;
}

// one carry pass over 32-bit limbs (each < 2^31): result reduced
VMPC_HD fe fe_carry32(const uint32_t h_in[FE_LIMBS]) {
    uint32_t h[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) h[i] = h_in[i];
    uint32_t c;
#pragma unroll
    for (int i = 0; i < FE_LIMBS - 1; i++) {
        if (i & 1) {
            c = h[i] >> 25;
            h[i] &= FE_MASK25;
        } else {
            c = h[i] >> 26;
            h[i] &= FE_MASK26;
        }
        h[i + 1] += c;
    }
    c = h[9] >> 25;
    h[9] &= FE_MASK25;
    h[0] += 19u * c;
    c = h[0] >> 26;
    h[0] &= FE_MASK26;
    h[1] += c;
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.v[i] = h[i];
    return r;
}

// carry pass over the 64-bit column sums of a product
VMPC_HD fe fe_carry64(uint64_t h[FE_LIMBS]) {
    uint64_t c;
#pragma unroll
    for (int i = 0; i < FE_LIMBS - 1; i++) {
        if (i & 1) {
            c = h[i] >> 25;
            h[i] &= FE_MASK25;
        } else {
            c = h[i] >> 26;
            h[i] &= FE_MASK26;
        }
        h[i + 1] += c;
    }
    c = h[9] >> 25;
    h[9] &= FE_MASK25;
    h[0] += 19u * c;
    c = h[0] >> 26;
    h[0] &= FE_MASK26;
    h[1] += c;
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.v[i] = (uint32_t)h[i];
    return r;
}

// limb-wise sum, NOT carried: only as an operand of fe_mul / fe_sqr / fe_sub / fe_add of
// reduced values (see the bounds at the top)
VMPC_HD fe fe_add_lazy(const fe &a, const fe &b) {
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}

// a - b + 2p limb-wise, NOT carried.  b must be reduced (limbs below the 2p limbs 2^27-38,
// 2^26-2, 2^27-2); the result's limbs are below a's + 2^27 (even) / 2^26 (odd).
VMPC_HD fe fe_sub_lazy(const fe &a, const fe &b) {
    fe r;
    r.v[0] = a.v[0] + 0x7ffffdau - b.v[0];          // 2 * (2^26 - 19)
#pragma unroll
    for (int i = 1; i < FE_LIMBS; i++)
        r.v[i] = a.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - b.v[i];   // 2*(2^25-1) / 2*(2^26-1)
    return r;
}
VMPC_HD fe fe_neg_lazy(const fe &a) { return fe_sub_lazy(fe_zero(), a); }

// r = a + b, reduced
VMPC_HD fe fe_add(const fe &a, const fe &b) {
    uint32_t h[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) h[i] = a.v[i] + b.v[i];
    return fe_carry32(h);
}

// r = a - b, reduced: a + 4p - b limb-wise (b's limbs may be up to 2^27 / 2^26), then one carry
VMPC_HD fe fe_sub(const fe &a, const fe &b) {
    uint32_t h[FE_LIMBS];
    h[0] = a.v[0] + 0xfffffb4u - b.v[0];          // 4 * (2^26 - 19)
#pragma unroll
    for (int i = 1; i < FE_LIMBS; i++)
        h[i] = a.v[i] + ((i & 1) ? 0x7fffffcu : 0xffffffcu) - b.v[i];   // 4*(2^25-1) / 4*(2^26-1)
    return fe_carry32(h);
}

VMPC_HD fe fe_neg(const fe &a) { return fe_sub(fe_zero(), a); }
VMPC_HD fe fe_dbl(const fe &a) { return fe_add(a, a); }

// r = a * b
VMPC_HD fe fe_mul(const fe &f, const fe &g) {
    uint32_t g19[FE_LIMBS], f2[FE_LIMBS];
#pragma unroll
    for (int j = 0; j < FE_LIMBS; j++) {
        g19[j] = 19u * g.v[j];
        f2[j] = 2u * f.v[j];
    }
    uint64_t h[FE_LIMBS];
#pragma unroll
    for (int k = 0; k < FE_LIMBS; k++) h[k] = 0;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
#pragma unroll
        for (int j = 0; j < FE_LIMBS; j++) {
            const int k = i + j;
            const bool wrap = k >= FE_LIMBS;
            const uint32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const uint32_t gj = wrap ? g19[j] : g.v[j];
            h[wrap ? k - FE_LIMBS : k] += (uint64_t)fi * gj;
        }
    }
    return fe_carry64(h);
}

// r = a^2: 55 products
VMPC_HD fe fe_sqr(const fe &f) {
    uint32_t f2[FE_LIMBS], f19[FE_LIMBS], f38[FE_LIMBS];
#pragma unroll
    for (int j = 0; j < FE_LIMBS; j++) {
        f2[j] = 2u * f.v[j];
        f19[j] = 19u * f.v[j];
        f38[j] = 38u * f.v[j];
    }
    uint64_t h[FE_LIMBS];
#pragma unroll
    for (int k = 0; k < FE_LIMBS; k++) h[k] = 0;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
#pragma unroll
        for (int j = i; j < FE_LIMBS; j++) {
            const int k = i + j;
            const bool wrap = k >= FE_LIMBS;
            const bool odd2 = (i & 1) && (j & 1);
            const uint32_t left = (i < j) ? f2[i] : f.v[i];
            const uint32_t right = wrap ? (odd2 ? f38[j] : f19[j]) : (odd2 ? f2[j] : f.v[j]);
            h[wrap ? k - FE_LIMBS : k] += (uint64_t)left * right;
        }
    }
    return fe_carry64(h);
}

// r = a * small (small < 2^32)
VMPC_HD fe fe_mul_u32(const fe &a, uint32_t s) {
    uint64_t h[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) h[i] = (uint64_t)a.v[i] * s;
    fe r = fe_carry64(h);          // h[0] may still hold a large 19*c term: one more pass
    uint32_t t[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) t[i] = r.v[i];
    return fe_carry32(t);
}

VMPC_HD fe fe_select(const fe &a, const fe &b, bool pick_b) {
    fe r;
    uint32_t m = 0u - (uint32_t)pick_b;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}

// ---- packed form: 8 x uint32 little-endian -----------------------------------------------------
struct fe8 {
    uint32_t w[8];
};

// any 256-bit integer -> limbs (bit 255 folds back as +19)
VMPC_HD fe fe_unpack(const uint32_t w[8]) {
    fe r;
    r.v[0] = w[0] & FE_MASK26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & FE_MASK25;
    r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & FE_MASK26;
    r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & FE_MASK25;
    r.v[4] = (w[3] >> 6) & FE_MASK26;
    r.v[5] = w[4] & FE_MASK25;
    r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & FE_MASK26;
    r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & FE_MASK25;
    r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & FE_MASK26;
    r.v[9] = (w[7] >> 6) & FE_MASK25;
    r.v[0] += 19u * (w[7] >> 31);
    return r;
}

// limbs -> canonical residue in [0, p) as 8 words
VMPC_HD fe8 fe_pack(const fe &a) {
    // two carry passes: every limb strictly inside its width, value < 2^255 + 19*2
    uint32_t t[FE_LIMBS];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) t[i] = a.v[i];
    fe r = fe_carry32(t);
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) t[i] = r.v[i];
    r = fe_carry32(t);
    // assemble the (at most 256-bit) integer
    uint64_t acc[8];
    acc[0] = (uint64_t)r.v[0] + ((uint64_t)r.v[1] << 26);
    acc[1] = ((uint64_t)r.v[2] << 19);
    acc[2] = ((uint64_t)r.v[3] << 13);
    acc[3] = ((uint64_t)r.v[4] << 6);
    acc[4] = (uint64_t)r.v[5] + ((uint64_t)r.v[6] << 25);
    acc[5] = ((uint64_t)r.v[7] << 19);
    acc[6] = ((uint64_t)r.v[8] << 12);
    acc[7] = ((uint64_t)r.v[9] << 6);
    uint32_t w[8];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += acc[i];
        w[i] = (uint32_t)c;
        c >>= 32;
    }
    // value < 2^256: subtract p while >= p (at most twice; value < 2p + small)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        uint32_t s[8];
        uint64_t d = 19;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d += w[i];
            s[i] = (uint32_t)d;
            d >>= 32;
        }
        uint32_t ge = (uint32_t)d | (s[7] >> 31);
        s[7] &= 0x7fffffffu;
        if (d) s[7] |= 0x80000000u;
        uint32_t m = 0u - (ge & 1u);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = (s[i] & m) | (w[i] & ~m);
    }
    fe8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.w[i] = w[i];
    return o;
}

// canonical limbs (every limb inside its width, value < p)
VMPC_HD fe fe_canon(const fe &a) {
    fe8 p = fe_pack(a);
    return fe_unpack(p.w);
}

VMPC_HD bool fe_is_zero(const fe &a) {
    fe8 c = fe_pack(a);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= c.w[i];
    return o == 0;
}

VMPC_HD bool fe_eq(const fe &a, const fe &b) { return fe_is_zero(fe_sub(a, b)); }

// w < p as a raw 256-bit integer (canonical-encoding check at the C-ABI boundary)
VMPC_HD bool fe8_is_canonical(const uint32_t w[8]) {
    if (w[7] >> 31) return false;
    if (w[7] != 0x7fffffffu) return true;
#pragma unroll
    for (int i = 6; i >= 1; i--)
        if (w[i] != 0xffffffffu) return true;
    return w[0] < 0xffffffedu;
}

VMPC_HD fe fe_sqr_n(fe a, int n) {
    for (int i = 0; i < n; i++) a = fe_sqr(a);
    return a;
}

// a^(p-2): the standard 254-squaring / 11-multiplication chain for 2^255-21.
VMPC_HD fe fe_inv(const fe &z) {
    fe z2 = fe_sqr(z);                        // 2
    fe z9 = fe_mul(fe_sqr_n(z2, 2), z);       // 9
    fe z11 = fe_mul(z9, z2);                  // 11
    fe z2_5_0 = fe_mul(fe_sqr(z11), z9);      // 2^5 - 1
    fe z2_10_0 = fe_mul(fe_sqr_n(z2_5_0, 5), z2_5_0);
    fe z2_20_0 = fe_mul(fe_sqr_n(z2_10_0, 10), z2_10_0);
    fe z2_40_0 = fe_mul(fe_sqr_n(z2_20_0, 20), z2_20_0);
    fe z2_50_0 = fe_mul(fe_sqr_n(z2_40_0, 10), z2_10_0);
    fe z2_100_0 = fe_mul(fe_sqr_n(z2_50_0, 50), z2_50_0);
    fe z2_200_0 = fe_mul(fe_sqr_n(z2_100_0, 100), z2_100_0);
    fe z2_250_0 = fe_mul(fe_sqr_n(z2_200_0, 50), z2_50_0);
    return fe_mul(fe_sqr_n(z2_250_0, 5), z11);  // 2^255 - 21
}

// ---- memory: packed 32-byte elements (public buffers) and raw limbs (workspaces) -----------------
VMPC_HD fe fe_load(const uint32_t *p) { return fe_unpack(p); }

VMPC_HD void fe_store(uint32_t *p, const fe &a) {
    fe8 c = fe_pack(a);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = c.w[i];
}

VMPC_HD fe fe_load_limbs(const uint32_t *p) {
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.v[i] = p[i];
    return r;
}

VMPC_HD void fe_store_limbs(uint32_t *p, const fe &a) {
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) p[i] = a.v[i];
}

// curve constants d = -121665/121666 and 2d (reduced limbs)
VMPC_HD fe fe_const_d() {
    fe r = {{0x35978a3u, 0x0d37284u, 0x3156ebdu, 0x06a0a0eu, 0x001c029u, 0x179e898u, 0x3a03cbbu,
             0x1ce7198u, 0x2e2b6ffu, 0x1480db3u}};
    return r;
}
VMPC_HD fe fe_const_d2() {
    fe r = {{0x2b2f159u, 0x1a6e509u, 0x22add7au, 0x0d4141du, 0x0038052u, 0x0f3d130u, 0x3407977u,
             0x19ce331u, 0x1c56dffu, 0x0901b67u}};
    return r;
}
// ---- shared helper of the Montgomery fields (sw256.h): 96-bit multiply-accumulate ----------------
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VMPC_NO_DEVICE_ASM)
#define VMPC_DEVICE_ASM 1
__device__ __forceinline__ void fe_mac96(uint64_t &acc, uint32_t &ovf, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf)
        : "v"(a), "v"(b)
        : "vcc");
}
#endif
