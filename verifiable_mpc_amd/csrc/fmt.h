// Decimal formatting of field elements / points, device side.
//
// The reference's Fiat-Shamir pre-image is str(input_list)
// (verifiable_mpc/ac20/pivot.py:131-136): every generator of the current round
// (compressed_pivot.py:52) and every coefficient of the linear form are stringified as
// decimal integers - O(N) big-int -> decimal conversions per round.  These helpers
// produce exactly that text on the device so only bytes ready for SHA-256 cross PCIe.
//   point  -> "[X, Y, Z]"   (un-normalised projective coordinates, unsigned decimals)
//   scalar -> signed decimal of the residue in (-l/2, l/2]
// [mpyc-recall: formats as restated in oracle/ed25519_ref.py pt_repr / scalar_repr]
// The recalled choices - bracket pair, signed or unsigned coordinates, signed or unsigned scalars - are RUNTIME
// settings (vmpc_repr_format, include/vmpc.h vmpc_set_reference_format): a mismatch with real MPyC found by
// scripts/check_against_mpyc.py is one call, not a kernel edit.
#pragma once
#include "fe25519.h"
#include "fr.h"

// v (256-bit) -> 9 base-10^9 chunks, least significant first.  Returns #decimal digits.
VMPC_HD int u256_to_chunks(const uint32_t v[8], uint32_t chunk[9]) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = v[i];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint64_t rem = 0;
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            uint64_t cur = (rem << 32) | t[i];
            uint64_t q = cur / 1000000000ull;
            rem = cur - q * 1000000000ull;
            t[i] = (uint32_t)q;
        }
        chunk[k] = (uint32_t)rem;
    }
    int top = 0;
#pragma unroll
    for (int k = 0; k < 9; k++)
        if (chunk[k]) top = k;
    uint32_t c = chunk[top];
    int d = 1;
    while (c >= 10) {
        c /= 10;
        d++;
    }
    return 9 * top + d;
}

VMPC_HD int u256_decimal_len(const uint32_t v[8]) {
    uint32_t chunk[9];
    return u256_to_chunks(v, chunk);
}

// writes the decimal digits of v at dst, returns the length
VMPC_HD int u256_write_decimal(const uint32_t v[8], char *dst) {
    uint32_t chunk[9];
    int len = u256_to_chunks(v, chunk);
    // digit position p (0 = least significant) lives in chunk[p / 9]
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint32_t c = chunk[k];
#pragma unroll
        for (int j = 0; j < 9; j++) {
            int pos = 9 * k + j;
            uint32_t q = c / 10;
            if (pos < len) dst[len - 1 - pos] = (char)('0' + (c - q * 10));
            c = q;
        }
    }
    return len;
}

// how a point is printed: bracket pair and whether a coordinate c > (p - 1) / 2 appears as -(p - c)
struct fmt_point_style {
    char open, close;
    int coord_signed;
};

// canonical coordinate c of GF(2^255 - 19): is it printed negative, and its magnitude
VMPC_HD bool fe8_signed_abs(const uint32_t c[8], uint32_t mag[8]) {
    // (p - 1) / 2 = 2^254 - 10: c > that  <=>  c >= 2^254 - 9
    bool big = (c[7] >> 30) != 0;                       // c >= 2^254 (c < p < 2^255)
    if (!big && c[7] == 0x3fffffffu) {
        bool all = true;
        for (int i = 1; i < 7; i++) all = all && c[i] == 0xffffffffu;
        big = all && c[0] >= 0xfffffff7u;               // 2^254 - 9 .. 2^254 - 1
    }
    if (!big) {
        for (int i = 0; i < 8; i++) mag[i] = c[i];
        return false;
    }
    const uint32_t pw[8] = {0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu,
                            0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};
    uint64_t borrow = 0;
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)pw[i] - c[i] - borrow;
        mag[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    return true;
}

VMPC_HD int coord_repr_len(const uint32_t c[8], int coord_signed) {
    if (!coord_signed) return u256_decimal_len(c);
    uint32_t mag[8];
    const bool neg = fe8_signed_abs(c, mag);
    return (neg ? 1 : 0) + u256_decimal_len(mag);
}

VMPC_HD int coord_repr_write(const uint32_t c[8], int coord_signed, char *dst) {
    if (!coord_signed) return u256_write_decimal(c, dst);
    uint32_t mag[8];
    const bool neg = fe8_signed_abs(c, mag);
    int o = 0;
    if (neg) dst[o++] = '-';
    return o + u256_write_decimal(mag, dst + o);
}

// "[X, Y, Z]" for canonical coordinates (brackets and coordinate signedness from the style)
VMPC_HD int proj_repr_len(const uint32_t X[8], const uint32_t Y[8], const uint32_t Z[8], const fmt_point_style &st) {
    return 6 + coord_repr_len(X, st.coord_signed) + coord_repr_len(Y, st.coord_signed) + coord_repr_len(Z, st.coord_signed);
}

VMPC_HD int proj_repr_write(const uint32_t X[8], const uint32_t Y[8], const uint32_t Z[8], const fmt_point_style &st,
                            char *dst) {
    int o = 0;
    dst[o++] = st.open;
    o += coord_repr_write(X, st.coord_signed, dst + o);
    dst[o++] = ',';
    dst[o++] = ' ';
    o += coord_repr_write(Y, st.coord_signed, dst + o);
    dst[o++] = ',';
    dst[o++] = ' ';
    o += coord_repr_write(Z, st.coord_signed, dst + o);
    dst[o++] = st.close;
    return o;
}

// signed decimal of a canonical residue mod l (or unsigned when is_signed == false)
VMPC_HD int fr_repr_len(const fr &a, bool is_signed) {
    fr mag = a;
    bool neg = is_signed ? fr_signed_abs(a, mag) : false;
    return (neg ? 1 : 0) + u256_decimal_len(mag.v);
}

VMPC_HD int fr_repr_write(const fr &a, bool is_signed, char *dst) {
    fr mag = a;
    bool neg = is_signed ? fr_signed_abs(a, mag) : false;
    int o = 0;
    if (neg) dst[o++] = '-';
    o += u256_write_decimal(mag.v, dst + o);
    return o;
}
