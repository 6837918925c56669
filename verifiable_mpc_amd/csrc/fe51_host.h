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
    const uint64_t c = (uint64_t)(t4 >> 51);
    r.v[4] = (uint64_t)t4 & MASK;
    r.v[0] += 19 * c;                       // limbs below 2^51 + 2^15 in: every t below 2^109, c below 2^58
    r.v[1] += r.v[0] >> 51;
    r.v[0] &= MASK;
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
}  // namespace fe51
