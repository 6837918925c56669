// GF(2^255 - 19) on the HOST, five 51-bit limbs with 128-bit products: the per-round host work of the prover
// (prover.hip: A_i, B_i to affine between two rounds - one inversion for the pair) sits on the critical path of
// every round, and the device-side 10-limb arithmetic compiled for the host takes ~15 us for that inversion;
// this form takes ~3.
#pragma once
#include <stdint.h>
#include <string.h>

namespace fe51 {
typedef unsigned __int128 u128;
struct el {
    uint64_t v[5];
};
static const uint64_t MASK = (1ull << 51) - 1;

// 32 little-endian bytes (any value below 2^256; bit 255 counts as 2^255 = 19 mod p)
static inline el from_bytes(const uint8_t s[32]) {
    uint64_t w[4];
    memcpy(w, s, 32);
    el r;
    r.v[0] = w[0] & MASK;
    r.v[1] = ((w[0] >> 51) | (w[1] << 13)) & MASK;
    r.v[2] = ((w[1] >> 38) | (w[2] << 26)) & MASK;
    r.v[3] = ((w[2] >> 25) | (w[3] << 39)) & MASK;
    r.v[4] = (w[3] >> 12) & MASK;
    r.v[0] += 19 * (w[3] >> 63);
    return r;
}

static inline el mul(const el &a, const el &b) {
    const uint64_t a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
    const uint64_t b0 = b.v[0], b1 = b.v[1], b2 = b.v[2], b3 = b.v[3], b4 = b.v[4];
    const uint64_t b1_19 = 19 * b1, b2_19 = 19 * b2, b3_19 = 19 * b3, b4_19 = 19 * b4;
    u128 t0 = (u128)a0 * b0 + (u128)a1 * b4_19 + (u128)a2 * b3_19 + (u128)a3 * b2_19 + (u128)a4 * b1_19;
    u128 t1 = (u128)a0 * b1 + (u128)a1 * b0 + (u128)a2 * b4_19 + (u128)a3 * b3_19 + (u128)a4 * b2_19;
    u128 t2 = (u128)a0 * b2 + (u128)a1 * b1 + (u128)a2 * b0 + (u128)a3 * b4_19 + (u128)a4 * b3_19;
    u128 t3 = (u128)a0 * b3 + (u128)a1 * b2 + (u128)a2 * b1 + (u128)a3 * b0 + (u128)a4 * b4_19;
    u128 t4 = (u128)a0 * b4 + (u128)a1 * b3 + (u128)a2 * b2 + (u128)a3 * b1 + (u128)a4 * b0;
    el r;
    t1 += (uint64_t)(t0 >> 51);
    r.v[0] = (uint64_t)t0 & MASK;
    t2 += (uint64_t)(t1 >> 51);
    r.v[1] = (uint64_t)t1 & MASK;
    t3 += (uint64_t)(t2 >> 51);
    r.v[2] = (uint64_t)t2 & MASK;
    t4 += (uint64_t)(t3 >> 51);
    r.v[3] = (uint64_t)t3 & MASK;
    const u128 c = t4 >> 51;                // (in 128 bits: operands may be sums of two reduced elements)
    r.v[4] = (uint64_t)t4 & MASK;
    const u128 f = (u128)r.v[0] + 19 * c;
    r.v[0] = (uint64_t)f & MASK;
    r.v[1] += (uint64_t)(f >> 51);
    return r;
}

static inline el sqr(const el &a) { return mul(a, a); }

static inline el sqr_n(el a, int n) {
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
}

// a^(p - 2) = a^(2^255 - 21): 254 squarings, 11 multiplications
static inline el inv(const el &z) {
    const el z2 = sqr(z);
    const el z9 = mul(z, sqr_n(z2, 2));
    const el z11 = mul(z2, z9);
    const el z_5_0 = mul(z9, sqr(z11));                        // 2^5 - 1
    const el z_10_0 = mul(sqr_n(z_5_0, 5), z_5_0);
    const el z_20_0 = mul(sqr_n(z_10_0, 10), z_10_0);
    const el z_40_0 = mul(sqr_n(z_20_0, 20), z_20_0);
    const el z_50_0 = mul(sqr_n(z_40_0, 10), z_10_0);
    const el z_100_0 = mul(sqr_n(z_50_0, 50), z_50_0);
    const el z_200_0 = mul(sqr_n(z_100_0, 100), z_100_0);
    const el z_250_0 = mul(sqr_n(z_200_0, 50), z_50_0);
    return mul(sqr_n(z_250_0, 5), z11);
}

// the canonical residue, 32 little-endian bytes
static inline void to_bytes(uint8_t out[32], const el &a) {
    uint64_t t[5] = {a.v[0], a.v[1], a.v[2], a.v[3], a.v[4]};
    for (int pass = 0; pass < 2; pass++) {                      // limbs below 2^51, value below 2 p
        t[1] += t[0] >> 51; t[0] &= MASK;
        t[2] += t[1] >> 51; t[1] &= MASK;
        t[3] += t[2] >> 51; t[2] &= MASK;
        t[4] += t[3] >> 51; t[3] &= MASK;
        t[0] += 19 * (t[4] >> 51); t[4] &= MASK;
    }
    // q = 1 iff value >= p: add 19 and see whether bit 255 carries
    uint64_t q = (t[0] + 19) >> 51;
    q = (t[1] + q) >> 51;
    q = (t[2] + q) >> 51;
    q = (t[3] + q) >> 51;
    q = (t[4] + q) >> 51;
    t[0] += 19 * q;
    t[1] += t[0] >> 51; t[0] &= MASK;
    t[2] += t[1] >> 51; t[1] &= MASK;
    t[3] += t[2] >> 51; t[2] &= MASK;
    t[4] += t[3] >> 51; t[3] &= MASK;
    t[4] &= MASK;
    uint64_t w[4];
    w[0] = t[0] | (t[1] << 51);
    w[1] = (t[1] >> 13) | (t[2] << 38);
    w[2] = (t[2] >> 26) | (t[3] << 25);
    w[3] = (t[3] >> 39) | (t[4] << 12);
    memcpy(out, w, 32);
}

static inline el add(const el &a, const el &b) {                // limbs below 2^52 + 2^16 out
    el r;
    for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
static inline el sub(const el &a, const el &b) {                // a + 4p - b, carried: limbs below 2^51 + 2^3 out
    static const uint64_t P4[5] = {4 * (MASK - 18), 4 * MASK, 4 * MASK, 4 * MASK, 4 * MASK};
    uint64_t t[5];
    for (int i = 0; i < 5; i++) t[i] = a.v[i] + P4[i] - b.v[i];  // b's limbs are below 2^53: no borrow
    el r;
    t[1] += t[0] >> 51; r.v[0] = t[0] & MASK;
    t[2] += t[1] >> 51; r.v[1] = t[1] & MASK;
    t[3] += t[2] >> 51; r.v[2] = t[2] & MASK;
    t[4] += t[3] >> 51; r.v[3] = t[3] & MASK;
    r.v[0] += 19 * (t[4] >> 51); r.v[4] = t[4] & MASK;
    return r;
}

// ---- the curve on the host: -x^2 + y^2 = 1 + d x^2 y^2, extended coordinates (add-2008-hwcd-3, dbl-2008-hwcd) ----
// for group elements only (no particular representative): A + c Q + c^2 B of a round's commitment fold
// (compressed_pivot.py:66), where the reference transcript hashes the NORMALISED result.
struct pt {
    el X, Y, Z, T;
};
static inline el one() { return el{{1, 0, 0, 0, 0}}; }
static inline el zero() { return el{{0, 0, 0, 0, 0}}; }
static inline el d2() {
    static const uint8_t D[32] = {0xa3, 0x78, 0x59, 0x13, 0xca, 0x4d, 0xeb, 0x75, 0xab, 0xd8, 0x41, 0x41, 0x4d, 0x0a, 0x70, 0x00,
                                  0x98, 0xe8, 0x79, 0x77, 0x79, 0x40, 0xc7, 0x8c, 0x73, 0xfe, 0x6f, 0x2b, 0xee, 0x6c, 0x03, 0x52};
    const el d = from_bytes(D);
    return sub(add(d, d), zero());
}
static inline pt pt_identity() { return pt{zero(), one(), one(), zero()}; }
static inline pt pt_from_affine(const uint8_t xy[64]) {
    pt r;
    r.X = from_bytes(xy);
    r.Y = from_bytes(xy + 32);
    r.Z = one();
    r.T = mul(r.X, r.Y);
    return r;
}
static inline pt pt_add(const pt &p, const pt &q, const el &dd) {
    const el A = mul(sub(p.Y, p.X), sub(q.Y, q.X));
    const el B = mul(add(p.Y, p.X), add(q.Y, q.X));
    const el C = mul(mul(p.T, dd), q.T);
    const el D = mul(add(p.Z, p.Z), q.Z);
    const el E = sub(B, A), F = sub(D, C), G = add(D, C), H = add(B, A);
    return pt{mul(E, F), mul(G, H), mul(F, G), mul(E, H)};
}
static inline pt pt_dbl(const pt &p) {
    const el A = sqr(p.X), B = sqr(p.Y), Z2 = sqr(p.Z);
    const el C = add(Z2, Z2), H = add(A, B);
    const el E = sub(H, sqr(add(p.X, p.Y)));
    const el G = sub(A, B);
    const el F = add(C, G);
    return pt{mul(E, F), mul(G, H), mul(F, G), mul(E, H)};
}
// s * p, s: 32 little-endian bytes (any 256-bit integer)
static inline pt pt_mul(const pt &p, const uint8_t s[32], const el &dd) {
    pt r = pt_identity();
    bool any = false;
    for (int bit = 255; bit >= 0; bit--) {
        if (any) r = pt_dbl(r);
        if ((s[bit >> 3] >> (bit & 7)) & 1) {
            r = any ? pt_add(r, p, dd) : p;
            any = true;
        }
    }
    return r;
}
static inline void pt_to_affine(uint8_t out[64], const pt &p) {
    const el zi = inv(p.Z);
    to_bytes(out, mul(p.X, zi));
    to_bytes(out + 32, mul(p.Y, zi));
}
}  // namespace fe51
