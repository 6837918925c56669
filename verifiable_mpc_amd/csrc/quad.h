// Quad-cooperative point arithmetic helpers (gfx950).
//
// A point operation has levels of 3-4 independent field multiplications.  In latency-bound
// kernels (the Horner chain of the MSM, the fold of short vectors) four adjacent lanes share
// one point: lane q computes the q-th product of a level and the results are exchanged with
// DPP quad_perm broadcasts (plain VALU moves, no LDS).  EXEC must be full within the quad.
#pragma once
#include "fe25519.h"

__device__ __forceinline__ fe quad_bcast(const fe &a, int src /*0..3, compile-time after unroll*/) {
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) {
        int v = (int)a.v[i];
        int o;
        switch (src) {
            case 0: o = __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true); break;
            case 1: o = __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true); break;
            case 2: o = __builtin_amdgcn_mov_dpp(v, 0xaa, 0xf, 0xf, true); break;
            default: o = __builtin_amdgcn_mov_dpp(v, 0xff, 0xf, 0xf, true); break;
        }
        r.v[i] = (uint32_t)o;
    }
    return r;
}

__device__ __forceinline__ fe fe_pick4(const fe &a0, const fe &a1, const fe &a2, const fe &a3, int q) {
    // branch-free: lanes of a quad take different operands in the same instruction stream
    const uint32_t m0 = 0u - (uint32_t)(q == 0), m1 = 0u - (uint32_t)(q == 1);
    const uint32_t m2 = 0u - (uint32_t)(q == 2), m3 = 0u - (uint32_t)(q == 3);
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++)
        r.v[i] = (a0.v[i] & m0) | (a1.v[i] & m1) | (a2.v[i] & m2) | (a3.v[i] & m3);
    return r;
}


// ---- lane-distributed form: lane q of a quad HOLDS coordinate q (X, Y, Z, T) ---------------------
// A point operation in this form never gathers all four coordinates into every lane: the second-level
// products leave X3, Y3, Z3, T3 in lanes 0..3, which is exactly where the next operation wants them.
template <int CTRL>
__device__ __forceinline__ fe quad_perm(const fe &a) {      // DPP quad_perm, CTRL = p0 | p1<<2 | p2<<4 | p3<<6
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++)
        r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], CTRL, 0xf, 0xf, true);
    return r;
}
__device__ __forceinline__ fe quad_sel(const fe &a, const fe &b, bool pick_b) {   // v_cndmask per limb
    fe r;
#pragma unroll
    for (int i = 0; i < FE_LIMBS; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
    return r;
}
