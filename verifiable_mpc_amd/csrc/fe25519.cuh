// GF(2^255-19) arithmetic for gfx950, 8 x 32-bit saturated limbs.
//
// Replaces the MPyC prime-field element arithmetic that every curve operation of the
// reference's hot path bottoms out in (verifiable_mpc/ac20/pivot.py:143-144 ->
// mpyc.fingroups / mpyc.finfields; SURVEY.md section 8a "EllipticCurve element type").
//
// Values are kept "loosely reduced": any 256-bit residue representative (0 <= v < 2^256).
// 2^256 = 38 (mod p), so a carry out of the top limb folds back as +38.  fe_canon()
// produces the unique representative in [0, p) and is applied before bytes leave the
// device or two elements are compared.
//
// The products are written as 32x32->64 multiply-adds ((uint64_t)a*b + c) which hipcc
// lowers to v_mad_u64_u32; there is no MFMA use (255-bit modular integers).
//
// All functions are VMPC_HD so the same source is unit-tested on the host
// (tests/native/host_math_test.cpp) against the Python oracle.
#pragma once
#include <stdint.h>

#ifndef VMPC_HD
#if defined(__HIPCC__) || defined(__CUDACC__)
#define VMPC_HD __host__ __device__ __forceinline__
#else
#define VMPC_HD inline
#endif
#endif

struct fe {
    uint32_t v[8];
};

VMPC_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}

VMPC_HD fe fe_one() {
    fe r = fe_zero();
    r.v[0] = 1;
    return r;
}

VMPC_HD fe fe_from_u32(uint32_t x) {
    fe r = fe_zero();
    r.v[0] = x;
    return r;
}

// r = a + b  (mod p, loosely reduced)
VMPC_HD fe fe_add(const fe &a, const fe &b) {
    fe r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // fold the carry twice (the second fold can only trigger on a tiny value)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        uint64_t t = (uint64_t)r.v[0] + 38u * (uint32_t)c;
        r.v[0] = (uint32_t)t;
        c = t >> 32;
#pragma unroll
        for (int i = 1; i < 8; i++) {
            c += r.v[i];
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}

// r = a - b  (mod p, loosely reduced)
VMPC_HD fe fe_sub(const fe &a, const fe &b) {
    fe r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;  // arithmetic shift: 0 or -1
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        int64_t t = (int64_t)r.v[0] - 38 * (-c);  // c is 0 or -1: subtract 38 on borrow
        r.v[0] = (uint32_t)t;
        c = t >> 32;
#pragma unroll
        for (int i = 1; i < 8; i++) {
            c += (int64_t)r.v[i];
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}

VMPC_HD fe fe_neg(const fe &a) { return fe_sub(fe_zero(), a); }

// 512-bit -> 256-bit: t[0..15] -> lo + 38*hi, then fold the small carry.
VMPC_HD fe fe_reduce512(const uint32_t t[16]) {
    fe r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)t[i + 8] * 38u + t[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // c < 39: fold twice
#pragma unroll
    for (int k = 0; k < 2; k++) {
        uint64_t u = (uint64_t)r.v[0] + 38u * (uint32_t)c;
        r.v[0] = (uint32_t)u;
        c = u >> 32;
#pragma unroll
        for (int i = 1; i < 8; i++) {
            c += r.v[i];
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}

// ---- device fast path -------------------------------------------------------------------
// On gfx950 the generic C forms below compile to 74 v_mad_u64_u32 + 71 64-bit adds + ~200
// v_mov per multiplication (zero-extended addends need register pairs).  The device path
// scans product columns with a 96-bit accumulator kept in (acc:64, ovf:32): each partial
// product is one v_mad_u64_u32 whose carry-out feeds a v_addc, i.e. 64 mads + 64 addc.
// Measured on MI355X (scripts/fe_bench.hip): 190-216 vs 150 G mul/s chip-wide.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VMPC_NO_DEVICE_ASM)
#define VMPC_DEVICE_ASM 1
__device__ __forceinline__ void fe_mac96(uint64_t &acc, uint32_t &ovf, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf)
        : "v"(a), "v"(b)
        : "vcc");
}
#endif

// r = a * b
VMPC_HD fe fe_mul(const fe &a, const fe &b) {
    uint32_t t[16];
#ifdef VMPC_DEVICE_ASM
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j >= 0 && j < 8) fe_mac96(acc, ovf, a.v[i], b.v[j]);
        }
        t[k] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ovf << 32);
        ovf = 0;
    }
    t[15] = (uint32_t)acc;
#else
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)a.v[0] * b.v[j];
        t[j] = (uint32_t)c;
        c >>= 32;
    }
    t[8] = (uint32_t)c;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[i] * b.v[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
#endif
    return fe_reduce512(t);
}

// r = a^2  (off-diagonal products computed once and doubled)
VMPC_HD fe fe_sqr(const fe &a) {
    uint32_t t[16];
#if defined(VMPC_DEVICE_ASM) && defined(VMPC_SQR_ASM)   // measured no faster than the C form: off
    // per column: off-diagonal products once, doubled as a 96-bit value, plus the square
    uint64_t acc = 0;
    uint32_t ovf = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
        uint64_t o = 0;
        uint32_t oo = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j > i && j < 8) fe_mac96(o, oo, a.v[i], a.v[j]);
        }
        // (acc, ovf) += 2 * (o, oo)
        oo = (oo << 1) | (uint32_t)(o >> 63);
        o <<= 1;
        uint64_t s = acc + o;
        ovf += oo + (uint32_t)(s < acc);
        acc = s;
        if ((k & 1) == 0) fe_mac96(acc, ovf, a.v[k >> 1], a.v[k >> 1]);
        t[k] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ovf << 32);
        ovf = 0;
    }
    t[15] = (uint32_t)acc;
#else
    uint64_t c;
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
    // off-diagonal: sum_{i<j} a_i a_j
#pragma unroll
    for (int i = 0; i < 7; i++) {
        c = 0;
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            c += (uint64_t)a.v[i] * a.v[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
    // double
    uint32_t top = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        uint32_t nt = t[i] >> 31;
        t[i] = (t[i] << 1) | top;
        top = nt;
    }
    // add diagonal squares
    c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t sq = (uint64_t)a.v[i] * a.v[i];
        c += (uint64_t)t[2 * i] + (uint32_t)sq;
        t[2 * i] = (uint32_t)c;
        c >>= 32;
        c += (uint64_t)t[2 * i + 1] + (uint32_t)(sq >> 32);
        t[2 * i + 1] = (uint32_t)c;
        c >>= 32;
    }
#endif
    return fe_reduce512(t);
}

// r = a * small (small < 2^32)
VMPC_HD fe fe_mul_u32(const fe &a, uint32_t s) {
    fe r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] * s;
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // c < 2^32: fold c*38 (up to 38 bits) into limbs 0..1
#pragma unroll
    for (int k = 0; k < 2; k++) {
        uint64_t u = c * 38u;  // first pass: < 2^38; second pass: c is 0/1
        uint64_t w = (uint64_t)r.v[0] + (uint32_t)u;
        r.v[0] = (uint32_t)w;
        w = (w >> 32) + r.v[1] + (u >> 32);
        r.v[1] = (uint32_t)w;
        c = w >> 32;
#pragma unroll
        for (int i = 2; i < 8; i++) {
            c += r.v[i];
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}

VMPC_HD fe fe_dbl(const fe &a) { return fe_add(a, a); }

// canonical representative in [0, p)
VMPC_HD fe fe_canon(const fe &a) {
    // p = 2^255 - 19.  a < 2^256 = 2p + 38, so at most two subtractions of p.
    fe r = a;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        // s = r - p = r + 19 - 2^255
        uint32_t s[8];
        uint64_t c = 19;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += r.v[i];
            s[i] = (uint32_t)c;
            c >>= 32;
        }
        // r >= p  <=>  (r + 19) >= 2^255  <=> bit 255 of (r+19) set or carry out
        uint32_t ge = (uint32_t)c | (s[7] >> 31);
        s[7] &= 0x7fffffffu;
        if (c) s[7] |= 0x80000000u;  // r + 19 >= 2^256: after removing 2^255 the bit stays
        uint32_t m = 0u - (ge & 1u);
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = (s[i] & m) | (r.v[i] & ~m);
    }
    return r;
}

VMPC_HD bool fe_is_zero(const fe &a) {
    fe c = fe_canon(a);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= c.v[i];
    return o == 0;
}

VMPC_HD bool fe_eq(const fe &a, const fe &b) { return fe_is_zero(fe_sub(a, b)); }

// a < p as a raw 256-bit integer (canonical-encoding check at the C-ABI boundary)
VMPC_HD bool fe_is_canonical(const fe &a) {
    if (a.v[7] >> 31) return false;
    if (a.v[7] != 0x7fffffffu) return true;
#pragma unroll
    for (int i = 6; i >= 1; i--)
        if (a.v[i] != 0xffffffffu) return true;
    return a.v[0] < 0xffffffedu;
}

VMPC_HD fe fe_select(const fe &a, const fe &b, bool pick_b) {
    fe r;
    uint32_t m = 0u - (uint32_t)pick_b;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
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

// ---- memory format: 32 bytes little-endian == 8 LE uint32 limbs --------------------
VMPC_HD fe fe_load(const uint32_t *p) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = p[i];
    return r;
}

VMPC_HD void fe_store(uint32_t *p, const fe &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = a.v[i];
}

// curve constants (little-endian limbs)
// d  = -121665/121666
#define VMPC_FE_D                                                                              \
    {                                                                                          \
        { 0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u,        \
          0x2b6ffe73u, 0x52036ceeu }                                                           \
    }
// 2d
#define VMPC_FE_D2                                                                             \
    {                                                                                          \
        { 0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u,        \
          0x56dffce7u, 0x2406d9dcu }                                                           \
    }

VMPC_HD fe fe_const_d() {
    fe r = VMPC_FE_D;
    return r;
}
VMPC_HD fe fe_const_d2() {
    fe r = VMPC_FE_D2;
    return r;
}
