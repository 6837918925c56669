// Device-side loads and stores of field elements and points in the layouts the kernels share:
// packed 32-byte canonical elements (public buffers), raw 10-limb elements (workspace, LDS),
// 128-byte niels lines (prepared points, fixed-base tables), extended points.
#pragma once
#include "fe25519.h"
#include "ge25519.h"

__device__ __forceinline__ void load_u32x8(uint32_t dst[8], const uint32_t *src) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    uint4 a = p[0], b = p[1];
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
}
// packed 32-byte canonical elements (public buffers)
__device__ __forceinline__ fe fe_ld8(const uint32_t *src) {
    uint32_t w[8];
    load_u32x8(w, src);
    return fe_unpack(w);
}
__device__ __forceinline__ void fe_st8(uint32_t *dst, const fe &a) {
    fe8 c = fe_pack(a);
    uint4 *p = reinterpret_cast<uint4 *>(dst);
    p[0] = make_uint4(c.w[0], c.w[1], c.w[2], c.w[3]);
    p[1] = make_uint4(c.w[4], c.w[5], c.w[6], c.w[7]);
}
// raw limbs (workspace buffers and LDS): 10 words = five 8-byte accesses
__device__ __forceinline__ fe fe_ld(const uint32_t *src) {
    const uint2 *p = reinterpret_cast<const uint2 *>(src);
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS / 2; i++) {
        uint2 v = p[i];
        r.v[2 * i] = v.x;
        r.v[2 * i + 1] = v.y;
    }
    return r;
}
__device__ __forceinline__ void fe_st(uint32_t *dst, const fe &a) {
    uint2 *p = reinterpret_cast<uint2 *>(dst);
#pragma unroll
    for (int i = 0; i < FE_LIMBS / 2; i++) p[i] = make_uint2(a.v[2 * i], a.v[2 * i + 1]);
}

#define EXT_WORDS (4 * FE_LIMBS)      // extended point in a workspace buffer / LDS
// niels entry: 30 limbs padded to one 128-byte line, moved with eight 16-byte accesses
#ifndef NIELS_WORDS
#define NIELS_WORDS 32
#endif

__device__ __forceinline__ ge_niels niels_ld_line(const uint32_t *src) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint4 v = p[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    ge_niels q;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
        q.ymx.v[i] = w[i];
        q.ypx.v[i] = w[FE_LIMBS + i];
        q.t2d.v[i] = w[2 * FE_LIMBS + i];
    }
    return q;
}
__device__ __forceinline__ void niels_st_line(uint32_t *dst, const ge_niels &q) {
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
        w[i] = q.ymx.v[i];
        w[FE_LIMBS + i] = q.ypx.v[i];
        w[2 * FE_LIMBS + i] = q.t2d.v[i];
    }
    w[30] = 0;
    w[31] = 0;
    uint4 *p = reinterpret_cast<uint4 *>(dst);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

__device__ __forceinline__ ge_ext ext_ld(const uint32_t *p) {
    ge_ext r;
    r.X = fe_ld(p);
    r.Y = fe_ld(p + FE_LIMBS);
    r.Z = fe_ld(p + 2 * FE_LIMBS);
    r.T = fe_ld(p + 3 * FE_LIMBS);
    return r;
}
__device__ __forceinline__ void ext_st(uint32_t *p, const ge_ext &a) {
    fe_st(p, a.X);
    fe_st(p + FE_LIMBS, a.Y);
    fe_st(p + 2 * FE_LIMBS, a.Z);
    fe_st(p + 3 * FE_LIMBS, a.T);
}
// packed 128-byte extended point X||Y||Z||T (public: partial sums exchanged between ranks)
__device__ __forceinline__ ge_ext ext_ld8(const uint32_t *p) {
    ge_ext r;
    r.X = fe_ld8(p);
    r.Y = fe_ld8(p + 8);
    r.Z = fe_ld8(p + 16);
    r.T = fe_ld8(p + 24);
    return r;
}
__device__ __forceinline__ void ext_st8(uint32_t *p, const ge_ext &a) {
    fe_st8(p, a.X);
    fe_st8(p + 8, a.Y);
    fe_st8(p + 16, a.Z);
    fe_st8(p + 24, a.T);
}
