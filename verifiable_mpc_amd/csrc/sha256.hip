// SHA-256 of fixed-size chunks of a device buffer, one lane per chunk.
//
// Used by the build's "compact" Fiat-Shamir transcript: the digests of
// the generator vector and of the linear form are two-level hashes whose 4096-byte leaves are
// independent, so they are computed where the data lives and only 32 bytes per leaf cross
// PCIe.  (The reference transcript, pivot.py:131-136, is one sequential SHA-256 over the
// whole text and cannot be split; it stays on the host's hashlib.)
#include "common.h"

#ifndef SHA_BLOCK
#define SHA_BLOCK 256
#endif

__device__ __constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

__device__ __forceinline__ void sha_compress(uint32_t st[8], uint32_t w[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + SHA_K[i] + w[i & 15];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__global__ void __launch_bounds__(SHA_BLOCK)
k_sha256_chunks(const uint8_t *__restrict__ data, size_t nbytes, size_t chunk, size_t n_chunks,
                uint32_t *__restrict__ out) {
    size_t ci = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= n_chunks) return;
    const uint8_t *p = data + ci * chunk;
    size_t len = nbytes - ci * chunk;
    if (len > chunk) len = chunk;
    uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                      0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint32_t w[16];
    size_t full = len / 64;
    const bool aligned = ((uintptr_t)p & 15) == 0;
    // (aligned: the NEXT block's 64 bytes are requested before this block is compressed - a lane walks its own chunk,
    // 64 lanes touch 64 different lines per request, and with one wave per SIMD nothing else hides that latency)
    uint4 nxt[4];
    if (aligned && full) {
        const uint4 *q4 = reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int i = 0; i < 4; i++) nxt[i] = q4[i];
    }
    for (size_t b = 0; b < full; b++) {
        const uint8_t *q = p + 64 * b;
        if (aligned) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 v = nxt[i];
                w[4 * i] = __builtin_bswap32(v.x);
                w[4 * i + 1] = __builtin_bswap32(v.y);
                w[4 * i + 2] = __builtin_bswap32(v.z);
                w[4 * i + 3] = __builtin_bswap32(v.w);
            }
            if (b + 1 < full) {
                const uint4 *q4 = reinterpret_cast<const uint4 *>(q + 64);
#pragma unroll
                for (int i = 0; i < 4; i++) nxt[i] = q4[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++)
                w[i] = ((uint32_t)q[4 * i] << 24) | ((uint32_t)q[4 * i + 1] << 16) |
                       ((uint32_t)q[4 * i + 2] << 8) | q[4 * i + 3];
        }
        sha_compress(st, w);
    }
    // tail + padding (one or two blocks).  A chunk that is a whole number of blocks - every 4096-byte leaf but a
    // buffer's last - ends with ONE constant block; the general case builds its words byte by byte from a function of
    // the position (a byte array indexed at run time became 56 000 instructions of lane-indexed register moves: as much
    // time as thirty blocks, for every lane).
    const size_t rem = len - 64 * full;
    const uint64_t bits = (uint64_t)len * 8;
    if (rem == 0) {
        w[0] = 0x80000000u;
#pragma unroll
        for (int i = 1; i < 14; i++) w[i] = 0;
        w[14] = (uint32_t)(bits >> 32);
        w[15] = (uint32_t)bits;
        sha_compress(st, w);
    } else {
        const int blocks = rem < 56 ? 1 : 2;
        const uint8_t *t = p + 64 * full;
        for (int b = 0; b < blocks; b++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                uint32_t word = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const size_t pos = (size_t)64 * b + 4 * i + k;                // position in the padded tail
                    uint32_t byte = 0;
                    if (pos < rem) byte = t[pos];
                    else if (pos == rem) byte = 0x80;
                    else if (pos >= (size_t)64 * blocks - 8) byte = (uint32_t)(bits >> (8 * ((size_t)64 * blocks - 1 - pos))) & 0xffu;
                    word = (word << 8) | byte;
                }
                w[i] = word;
            }
            sha_compress(st, w);
        }
    }
    for (int i = 0; i < 8; i++) out[8 * ci + i] = __builtin_bswap32(st[i]);   // big-endian digest bytes
}

extern "C" int vmpc_sha256_chunks_dev(vmpc_ctx *ctx, const void *data, size_t nbytes, size_t chunk_bytes,
                                      void *out_digests) {
    if (!ctx || !chunk_bytes || (nbytes && (!data || !out_digests))) return VMPC_E_INVAL;
    if (nbytes == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    size_t n_chunks = (nbytes + chunk_bytes - 1) / chunk_bytes;
    vmpc_stage_scope s(ctx, "sha256_chunks");
    k_sha256_chunks<<<(unsigned)((n_chunks + SHA_BLOCK - 1) / SHA_BLOCK), SHA_BLOCK, 0, ctx->stream>>>(
        (const uint8_t *)data, nbytes, chunk_bytes, n_chunks, (uint32_t *)out_digests);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}
