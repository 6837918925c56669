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

// ---- point operations in the lane-distributed form ------------------------------------------------------
// Each doubling / addition has two levels of four independent field multiplications: lane q of the quad does the
// q-th product of each level.  Second level of both operations: lane 0: E*F = X3, lane 1: G*H = Y3, lane 2:
// G*F = Z3, lane 3: E*H = T3 - two-way operand selections, and the products land where the next operation reads
// them.  Sums stay lazy as in ge25519.h (only F is carried).
__device__ __forceinline__ fe quadD_level2(const fe &E, const fe &F, const fe &G, const fe &H, int q) {
    fe u = quad_sel(G, E, q == 0 || q == 3);
    fe v = quad_sel(H, F, (q & 1) == 0);
    return fe_mul(u, v);
}

__device__ __forceinline__ fe quadD_dbl(const fe &P, int q) {
    fe x = quad_perm<0x00>(P), y = quad_perm<0x55>(P);
    fe in = quad_sel(P, fe_add_lazy(x, y), q == 3);          // X, Y, Z, X+Y
    fe sq = fe_sqr(in);
    fe A = quad_perm<0x00>(sq), B = quad_perm<0x55>(sq), C = quad_perm<0xaa>(sq), S = quad_perm<0xff>(sq);
    fe H = fe_add_lazy(A, B);
    fe E = fe_sub_lazy(H, S);
    fe G = fe_sub_lazy(A, B);
    fe F = fe_add(fe_add_lazy(C, C), G);                      // carried
    return quadD_level2(E, F, G, H, q);
}

// P += r, lane q given its own first-level partner v_q of r = (Y-X, Y+X, 2d*T, 2*Z) (all reduced)
__device__ __forceinline__ fe quadD_add_cached(const fe &P, const fe &vq, int q) {
    fe x = quad_perm<0x00>(P);
    fe t = quad_perm<0xb5>(P);                                // lanes: Y, Y, T, Z
    fe u = quad_sel(quad_sel(t, fe_add_lazy(t, x), q == 1), fe_sub_lazy(t, x), q == 0);
    fe prod = fe_mul(u, vq);                                  // A, B, C, D
    fe A = quad_perm<0x00>(prod), B = quad_perm<0x55>(prod), C = quad_perm<0xaa>(prod), D = quad_perm<0xff>(prod);
    fe E = fe_sub_lazy(B, A);
    fe H = fe_add_lazy(B, A);
    fe F = fe_sub(D, C);                                      // carried
    fe G = fe_add_lazy(D, C);
    return quadD_level2(E, F, G, H, q);
}

// the cached form (Y-X, Y+X, 2d*T, 2*Z) of a lane-distributed point, lane q keeping its own entry
__device__ __forceinline__ fe quadD_to_cached(const fe &P, int q) {
    const fe x = quad_perm<0x00>(P), y = quad_perm<0x55>(P);
    const fe s = quad_perm<0xb4>(P);                           // lanes: X, Y, T, Z
    const fe m = fe_mul(s, fe_const_d2());
    fe r = quad_sel(fe_sub(y, x), fe_add(y, x), q == 1);
    r = quad_sel(r, m, q == 2);
    return quad_sel(r, fe_dbl(s), q == 3);
}
