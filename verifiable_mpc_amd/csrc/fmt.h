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

// "[X, Y, Z]" for canonical coordinates
VMPC_HD int proj_repr_len(const uint32_t X[8], const uint32_t Y[8], const uint32_t Z[8]) {
    return 6 + u256_decimal_len(X) + u256_decimal_len(Y) + u256_decimal_len(Z);
}

VMPC_HD int proj_repr_write(const uint32_t X[8], const uint32_t Y[8], const uint32_t Z[8],
                            char *dst) {
    int o = 0;
    dst[o++] = '[';
    o += u256_write_decimal(X, dst + o);
    dst[o++] = ',';
    dst[o++] = ' ';
    o += u256_write_decimal(Y, dst + o);
    dst[o++] = ',';
    dst[o++] = ' ';
    o += u256_write_decimal(Z, dst + o);
    dst[o++] = ']';
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
