// Scalar field GF(l), l = 2^252 + 27742317777372353535851937790883648493 (Ed25519 group
// order), 8 x 32-bit limbs, canonical residues in memory (32 bytes LE).
//
// Replaces the MPyC GF(l) element arithmetic of the scalar side of Protocol 4/5:
//   z' = z_l + c*z_r, L' = c*L_l + L_r   verifiable_mpc/ac20/compressed_pivot.py:70-76
//   z  = c0*x + r, L~ = (L||0)*c1        compressed_pivot.py:134,141
//   L(z) = sum coeffs[i]*values[i]       verifiable_mpc/ac20/pivot.py:84-92
#pragma once
#include <stdint.h>
#include "fe25519.h"  // VMPC_HD

struct fr {
    uint32_t v[8];
};

#define VMPC_FR_L                                                                              \
    { 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u,            \
      0x00000000u, 0x10000000u }
// mu = floor(2^512 / l), 9 limbs
#define VMPC_FR_MU                                                                             \
    { 0x0a2c131bu, 0xed9ce5a3u, 0x086329a7u, 0x2106215du, 0xffffffebu, 0xffffffffu,            \
      0xffffffffu, 0xffffffffu, 0x0000000fu }

VMPC_HD fr fr_zero() {
    fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}

VMPC_HD fr fr_load(const uint32_t *p) {
    fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = p[i];
    return r;
}

VMPC_HD void fr_store(uint32_t *p, const fr &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = a.v[i];
}

// a >= l ?
VMPC_HD bool fr_geq_l(const uint32_t a[8]) {
    const uint32_t L[8] = VMPC_FR_L;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a[i] > L[i]) return true;
        if (a[i] < L[i]) return false;
    }
    return true;
}

VMPC_HD bool fr_is_canonical(const fr &a) { return !fr_geq_l(a.v); }

VMPC_HD bool fr_is_zero(const fr &a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

// r = a - l if a >= l (a < 2l)
VMPC_HD fr fr_cond_sub_l(const fr &a) {
    const uint32_t L[8] = VMPC_FR_L;
    fr s;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)L[i];
        s.v[i] = (uint32_t)c;
        c >>= 32;
    }
    uint32_t m = (uint32_t)c;  // all ones if borrow (a < l): keep a
    fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (a.v[i] & m) | (s.v[i] & ~m);
    return r;
}

VMPC_HD fr fr_add(const fr &a, const fr &b) {
    fr r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return fr_cond_sub_l(r);  // a, b < l < 2^253: no carry out of 256 bits
}

VMPC_HD fr fr_sub(const fr &a, const fr &b) {
    const uint32_t L[8] = VMPC_FR_L;
    fr r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    uint32_t m = (uint32_t)c;  // borrow: add l back
    uint64_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d += (uint64_t)r.v[i] + (L[i] & m);
        r.v[i] = (uint32_t)d;
        d >>= 32;
    }
    return r;
}

VMPC_HD fr fr_neg(const fr &a) { return fr_sub(fr_zero(), a); }

// Barrett reduction of a 512-bit value (HAC 14.42 with b = 2^32, k = 8).
VMPC_HD fr fr_reduce512(const uint32_t x[16]) {
    const uint32_t L[8] = VMPC_FR_L;
    const uint32_t MU[9] = VMPC_FR_MU;
    // q1 = x >> 224 (9 limbs: x[7..15]); q2 = q1 * mu (18 limbs); q3 = q2 >> 288 (9 limbs)
    uint32_t q2[18];
#pragma unroll
    for (int i = 0; i < 18; i++) q2[i] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            c += (uint64_t)x[7 + i] * MU[j] + q2[i + j];
            q2[i + j] = (uint32_t)c;
            c >>= 32;
        }
        q2[i + 9] = (uint32_t)c;
    }
    // r2 = (q3 * l) mod 2^288, q3 = q2[9..17]
    uint32_t r2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) r2[i] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (i + j < 9) {
                c += (uint64_t)q2[9 + i] * L[j] + r2[i + j];
                r2[i + j] = (uint32_t)c;
                c >>= 32;
            }
        }
        if (i + 8 < 9) r2[i + 8] = (uint32_t)c;
    }
    // r = (x mod 2^288) - r2 (mod 2^288); 0 <= r < 3l
    uint32_t r[9];
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += (int64_t)x[i] - (int64_t)r2[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    // at most two subtractions of l (9-limb compare: r[8] may be nonzero only transiently)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        uint32_t s[9];
        int64_t b = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            b += (int64_t)r[i] - (int64_t)(i < 8 ? L[i] : 0u);
            s[i] = (uint32_t)b;
            b >>= 32;
        }
        uint32_t m = (uint32_t)b;  // borrow: keep r
#pragma unroll
        for (int i = 0; i < 9; i++) r[i] = (r[i] & m) | (s[i] & ~m);
    }
    fr out;
#pragma unroll
    for (int i = 0; i < 8; i++) out.v[i] = r[i];
    return out;
}

VMPC_HD void fr_mul_wide(uint32_t t[16], const fr &a, const fr &b) {
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
}

VMPC_HD fr fr_mul(const fr &a, const fr &b) {
    uint32_t t[16];
    fr_mul_wide(t, a, b);
    return fr_reduce512(t);
}

// reduce an arbitrary 256-bit value (e.g. a SHA-256 digest read as LE integer)
VMPC_HD fr fr_from_u256(const uint32_t a[8]) {
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t[i] = a[i];
        t[i + 8] = 0;
    }
    return fr_reduce512(t);
}

// signed residue split used by the reference's `_int` (pivot.py:119-128) on a signed
// GF(l): returns true and |a| = l - a when a > l/2  [mpyc-recall: GF() is signed]
VMPC_HD bool fr_signed_abs(const fr &a, fr &mag) {
    // l/2 = (l-1)/2 ; a > (l-1)/2  <=>  2a > l - 1  <=> 2a >= l
    uint32_t d[8];
    uint32_t top = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d[i] = (a.v[i] << 1) | top;
        top = a.v[i] >> 31;
    }
    bool neg = fr_geq_l(d);  // a < 2^253 so no overflow
    fr n = fr_neg(a);
    mag = neg ? n : a;
    return neg;
}
