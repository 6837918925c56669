// Element-wise kernels that REPLAY the reference's group-operation sequences
// (projective add-2008-bbjlp / dbl-2008-bbjlp, right-to-left `repeat`) so that the
// un-normalised coordinates entering the reference's Fiat-Shamir pre-image are
// reproduced bit for bit (see ge25519.h):
//   vmpc_fold_dev         g'_i = (g_l[i] ** c) * g_r[i]     compressed_pivot.py:64 / :178
//   vmpc_repeat_dev       base ** r_i, g[i] ** x_i          circuit_sat_r1cs.py:64-70,81; pivot.py:143
//   vmpc_tree_reduce_dev  pivot.list_mul                    pivot.py:26-28
//   vmpc_normalize_dev    .normalize() over a vector        compressed_pivot.py:52,118
// One lane per element; 96-B / 64-B elements are moved with 16-B accesses.  The fold has a
// wave-uniform scalar (no divergence); repeat predicates per lane.
#include <stdlib.h>

#include "common.h"
#include "fe25519.h"
#include "fr.h"
#include "ge25519.h"
#include "quad.h"

#define EX_BLOCK 256
// vectors up to this many elements use four lanes per element (latency-bound regime)
#define EX_FOLD_QUAD_MAX (16 * 1024)

// public buffers hold packed 32-byte canonical elements: two 16-byte accesses per element
__device__ __forceinline__ fe ex_fe_ld(const uint32_t *src) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    uint4 a = p[0], b = p[1];
    uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fe_unpack(w);
}
__device__ __forceinline__ void ex_fe_st(uint32_t *dst, const fe &a) {
    fe8 c = fe_pack(a);
    uint4 *p = reinterpret_cast<uint4 *>(dst);
    p[0] = make_uint4(c.w[0], c.w[1], c.w[2], c.w[3]);
    p[1] = make_uint4(c.w[4], c.w[5], c.w[6], c.w[7]);
}
__device__ __forceinline__ ge_proj ex_load_point(const uint32_t *base, size_t i, bool affine) {
    ge_proj p;
    if (affine) {
        p.X = ex_fe_ld(base + 16 * i);
        p.Y = ex_fe_ld(base + 16 * i + 8);
        p.Z = fe_one();
    } else {
        p.X = ex_fe_ld(base + 24 * i);
        p.Y = ex_fe_ld(base + 24 * i + 8);
        p.Z = ex_fe_ld(base + 24 * i + 16);
    }
    return p;
}
__device__ __forceinline__ void ex_store_point(const ge_proj &r, size_t i, uint32_t *out_proj,
                                               uint32_t *out_aff) {
    if (out_proj) {      // ex_fe_st writes the canonical residue
        ex_fe_st(out_proj + 24 * i, r.X);
        ex_fe_st(out_proj + 24 * i + 8, r.Y);
        ex_fe_st(out_proj + 24 * i + 16, r.Z);
    }
    if (out_aff) {
        ge_aff a = ge_proj_to_affine(r);
        ex_fe_st(out_aff + 16 * i, a.x);
        ex_fe_st(out_aff + 16 * i + 8, a.y);
    }
}

struct u256_arg {
    uint32_t v[8];
};

static inline unsigned ex_grid(size_t n) { return (unsigned)((n + EX_BLOCK - 1) / EX_BLOCK); }

// ---- fold ---------------------------------------------------------------------------------
#ifndef EX_FOLD_WAVES
#define EX_FOLD_WAVES 2
#endif
__global__ void __launch_bounds__(EX_BLOCK, EX_FOLD_WAVES)
k_fold(const uint32_t *__restrict__ gl, const uint32_t *__restrict__ gr, int in_affine, u256_arg c,
       size_t half, uint32_t *__restrict__ out_proj, uint32_t *__restrict__ out_aff) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    ge_proj r = ge_proj_repeat(ex_load_point(gl, i, in_affine != 0), c.v);   // g_l ** c
    r = ge_proj_add(r, ex_load_point(gr, i, in_affine != 0));                // * g_r (loaded late)
    ex_store_point(r, i, out_proj, out_aff);
}

// ---- fold, quad-cooperative (short vectors) ---------------------------------------------------
// Below ~2^16 elements the one-lane-per-element fold leaves the chip idle and costs the full
// ~3300-multiplication dependency chain (2.1 ms).  Here four lanes share an element and run
// the SAME formulas level by level (dbl-2008-bbjlp: 4 squarings | 3 products; add-2008-bbjlp:
// 4 | 4 | 1 | 3), so the chain is ~1000 multiplications long.  Field arithmetic is exact
// and commutative/associative, so the (X:Y:Z) produced are bit-identical to k_fold's.
// (sums stay lazy - uncarried - wherever fe_mul's operand bounds allow, exactly as in ge_proj_dbl / ge_proj_add,
// ge25519.h: a carry pass is ~25 instructions and the formulas have seven sums per operation)
__device__ __forceinline__ void quad_proj_dbl(ge_proj &p, int q) {
    fe in = fe_pick4(p.X, p.Y, p.Z, fe_add_lazy(p.X, p.Y), q);
    fe sq = fe_sqr(in);
    fe C = quad_bcast(sq, 0), D = quad_bcast(sq, 1), H = quad_bcast(sq, 2), B = quad_bcast(sq, 3);
    fe F = fe_sub_lazy(D, C);                                 // E + D with E = -C;  < 2^27.6
    fe J = fe_sub(F, fe_add_lazy(H, H));                      // carried
    fe nCD = fe_neg(fe_add_lazy(C, D));                       // E - D, carried
    fe u = fe_pick4(fe_sub_lazy(fe_sub_lazy(B, C), D), F, F, F, q);      // B - C - D < 2^28.4 against a reduced J
    fe v = fe_pick4(J, nCD, J, J, q);
    fe prod = fe_mul(u, v);
    p.X = quad_bcast(prod, 0);
    p.Y = quad_bcast(prod, 1);
    p.Z = quad_bcast(prod, 2);
}

// add-2008-bbjlp in FOUR levels (4 | 4 | 1 | 3 products; the formula's own order has five: 4 | 2 | 1 | 3 | 2): A times
// the two brackets of X3, Y3 moves up to the level of B = A^2 and C D.  Products are exact mod p, so the residues -
// and the canonical limbs that are stored - do not depend on that order.
__device__ __forceinline__ ge_proj quad_proj_add(const ge_proj &p, const ge_proj &r, int q) {
    // level 1: A = Z1 Z2, C = X1 X2, D = Y1 Y2, S = (X1+Y1)(X2+Y2)
    fe l1 = fe_mul(fe_pick4(p.Z, p.X, p.Y, fe_add_lazy(p.X, p.Y), q),
                   fe_pick4(r.Z, r.X, r.Y, fe_add_lazy(r.X, r.Y), q));
    fe A = quad_bcast(l1, 0), C = quad_bcast(l1, 1), D = quad_bcast(l1, 2), S = quad_bcast(l1, 3);
    // level 2: B = A^2, U = C D, A (S - C - D), A (D + C)
    fe DC = fe_add_lazy(D, C);                                // < 2^27.1
    fe l2 = fe_mul(fe_pick4(A, C, A, A, q), fe_pick4(A, D, fe_sub(S, DC), DC, q));       // (S - D - C carried)
    fe B = quad_bcast(l2, 0), U = quad_bcast(l2, 1), AS = quad_bcast(l2, 2), AC = quad_bcast(l2, 3);
    // level 3 (every lane): E = d * C * D
    fe E = fe_mul(fe_const_d(), U);
    fe F = fe_sub_lazy(B, E), G = fe_add_lazy(B, E);          // < 2^27.6, < 2^27.1
    // level 4: X3 = F * A (S - C - D), Y3 = G * A (D + C), Z3 = F G   (2^27.6 * 2^27.1 as in ge_proj_add)
    fe l4 = fe_mul(fe_pick4(F, G, F, F, q), fe_pick4(AS, AC, G, G, q));
    ge_proj o;
    o.X = quad_bcast(l4, 0);
    o.Y = quad_bcast(l4, 1);
    o.Z = quad_bcast(l4, 2);
    return o;
}

__global__ void __launch_bounds__(EX_BLOCK)
k_fold_quad(const uint32_t *__restrict__ gl, const uint32_t *__restrict__ gr, int in_affine, u256_arg c,
            size_t half, uint32_t *__restrict__ out_proj, uint32_t *__restrict__ out_aff) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t i = t >> 2;
    const int q = (int)(t & 3);
    const bool live = i < half;
    if (!live) i = half - 1;                 // keep the quad's EXEC full; result discarded
    ge_proj a = ex_load_point(gl, i, in_affine != 0);
    ge_proj b = ex_load_point(gr, i, in_affine != 0);
    // (g_l ** c): right-to-left binary, same sequence as ge_proj_repeat (c is wave-uniform)
    ge_proj r;
    int bl = u256_bit_length(c.v);
    if (bl == 0) {
        r = ge_proj_identity();
    } else {
        ge_proj d = a;
        ge_proj acc = ge_proj_identity();
#pragma unroll 1
        for (int bit = 0; bit < bl - 1; bit++) {
            if ((c.v[bit >> 5] >> (bit & 31)) & 1u) acc = quad_proj_add(acc, d, q);
            quad_proj_dbl(d, q);
        }
        r = quad_proj_add(acc, d, q);
    }
    r = quad_proj_add(r, b, q);              // * g_r
    if (live && q == 0) ex_store_point(r, i, out_proj, out_aff);
}

// ---- fold, two waves per 16 elements (the shortest vectors) --------------------------------------
// In the right-to-left ladder the doublings d_{j+1} = 2 d_j never wait for the additions acc += d_j - k_fold_quad runs
// both chains in one instruction stream and pays 2 + 4 levels of multiplications per set bit.  Here a workgroup is two
// waves over the same 16 elements (a quad per element in each): wave 0 walks the doubling chain and leaves d_j in LDS,
// FP_K of them at a time; wave 1, one batch behind, adds the d_j of the set bits - the SAME operations on the same
// operands in the same order, so the (X:Y:Z) are k_fold's bit for bit; the two chains run on two SIMDs side by side
// (0.78 -> ~0.4 ms for a full-length scalar, whatever the vector's length up to one wave per SIMD).
#define FP_K 8                 // doublings per batch
#define FP_ELEMS 16            // elements per workgroup
#define FP_MAX_DEFAULT (8 * 1024)

// LDS: [buffer][k][16-byte piece 0..7][element] - a piece index across the 16 elements is 256 contiguous bytes
__device__ __forceinline__ uint4 *fp_slot(uint4 *ring, int buf, int k, int piece, int elem) {
    return ring + (((buf * FP_K + k) * 8 + piece) * FP_ELEMS + elem);
}

// PAIRS: (doubling wave, adding wave) pairs per workgroup.  One pair per workgroup spreads a short vector over the most
// CUs; at 8192 elements that is two workgroups per CU and the dispatcher does not keep their four waves on four SIMDs
// (858 us against 520 for 4096 elements) - two pairs per workgroup, one workgroup per CU, does.
template <int PAIRS>
__global__ void __launch_bounds__(128 * PAIRS)
k_fold_pipe(const uint32_t *__restrict__ gl, const uint32_t *__restrict__ gr, int in_affine, u256_arg c,
            size_t half, uint32_t *__restrict__ out_proj, uint32_t *__restrict__ out_aff) {
    __shared__ uint4 ring_all[PAIRS * 2 * FP_K * 8 * FP_ELEMS];            // 32 KiB per pair
    const int wave = (threadIdx.x >> 6) / PAIRS, pair = (threadIdx.x >> 6) % PAIRS, lane = threadIdx.x & 63;
    uint4 *ring = ring_all + pair * (2 * FP_K * 8 * FP_ELEMS);
    const int elem = lane >> 2, q = lane & 3;
    size_t i = ((size_t)blockIdx.x * PAIRS + pair) * FP_ELEMS + elem;
    const bool live = i < half;
    if (!live) i = half - 1;
    const int bl = u256_bit_length(c.v);
    const int batches = (bl + FP_K - 1) / FP_K;
    ge_proj d, acc = ge_proj_identity();
    if (wave == 0) d = ex_load_point(gl, i, in_affine != 0);
#pragma unroll 1
    for (int t = 0; t <= batches; t++) {
        if (wave == 0) {
            if (t < batches) {
#pragma unroll 1
                for (int k = 0; k < FP_K; k++) {
                    if (t * FP_K + k >= bl) break;
                    // lanes 0, 1, 2 of the quad leave X, Y, Z (10 limbs each; pieces 0-2, 3-5 and 6-7 + the 8 spare
                    // bytes: X in pieces 0..2 words 0..9 ... laid out as 30 consecutive words + 2 of padding)
                    uint32_t w[32];
#pragma unroll
                    for (int l = 0; l < FE_LIMBS; l++) {
                        w[l] = d.X.v[l];
                        w[FE_LIMBS + l] = d.Y.v[l];
                        w[2 * FE_LIMBS + l] = d.Z.v[l];
                    }
                    w[30] = w[31] = 0;
                    // (every lane of the quad holds the whole point: lane q stores pieces 2q and 2q + 1)
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        uint4 v;
                        switch (q) {
                            case 0: v = make_uint4(w[0 + 4 * h], w[1 + 4 * h], w[2 + 4 * h], w[3 + 4 * h]); break;
                            case 1: v = make_uint4(w[8 + 4 * h], w[9 + 4 * h], w[10 + 4 * h], w[11 + 4 * h]); break;
                            case 2: v = make_uint4(w[16 + 4 * h], w[17 + 4 * h], w[18 + 4 * h], w[19 + 4 * h]); break;
                            default: v = make_uint4(w[24 + 4 * h], w[25 + 4 * h], w[26 + 4 * h], w[27 + 4 * h]); break;
                        }
                        *fp_slot(ring, t & 1, k, 2 * q + h, elem) = v;
                    }
                    quad_proj_dbl(d, q);
                }
            }
        } else if (t >= 1) {
            const int tb = t - 1;
#pragma unroll 1
            for (int k = 0; k < FP_K; k++) {
                const int bit = tb * FP_K + k;
                if (bit >= bl) break;
                if (!((c.v[bit >> 5] >> (bit & 31)) & 1u)) continue;
                uint32_t w[32];
#pragma unroll
                for (int piece = 0; piece < 8; piece++) {
                    const uint4 v = *fp_slot(ring, tb & 1, k, piece, elem);
                    w[4 * piece] = v.x;
                    w[4 * piece + 1] = v.y;
                    w[4 * piece + 2] = v.z;
                    w[4 * piece + 3] = v.w;
                }
                ge_proj dj;
#pragma unroll
                for (int l = 0; l < FE_LIMBS; l++) {
                    dj.X.v[l] = w[l];
                    dj.Y.v[l] = w[FE_LIMBS + l];
                    dj.Z.v[l] = w[2 * FE_LIMBS + l];
                }
                acc = quad_proj_add(acc, dj, q);
            }
        }
        __syncthreads();
    }
    if (wave == 1) {
        ge_proj r = quad_proj_add(acc, ex_load_point(gr, i, in_affine != 0), q);      // * g_r
        if (live && q == 0) ex_store_point(r, i, out_proj, out_aff);
    }
}

// ---- repeat -------------------------------------------------------------------------------
__global__ void __launch_bounds__(EX_BLOCK)
k_repeat(const uint32_t *__restrict__ bases, size_t n_bases, int bases_affine,
         const uint32_t *__restrict__ scalars, size_t n, int signed_scalars,
         uint32_t *__restrict__ out_proj, uint32_t *__restrict__ out_aff) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ge_proj a = ex_load_point(bases, n_bases == 1 ? 0 : i, bases_affine != 0);
    fr s;
    {
        const uint4 *p = reinterpret_cast<const uint4 *>(scalars + 8 * i);
        uint4 x = p[0], y = p[1];
        s.v[0] = x.x; s.v[1] = x.y; s.v[2] = x.z; s.v[3] = x.w;
        s.v[4] = y.x; s.v[5] = y.y; s.v[6] = y.z; s.v[7] = y.w;
    }
    if (signed_scalars == 2) {
        // sign-magnitude: |n| < 2^255 in bits 0..254, bit 255 set for n < 0 (exponents the caller has
        // already converted with pivot._int: plain Python ints are not residues and may exceed l)
        if (s.v[7] >> 31) a = ge_proj_neg(a);
        s.v[7] &= 0x7fffffffu;
    } else if (signed_scalars) {
        // pivot._int on a signed field element: residues above l/2 are negative ints, and
        // `a ** n` with n < 0 inverts the base first (oracle pt_repeat)
        fr mag;
        bool neg = fr_signed_abs(s, mag);
        if (neg) a = ge_proj_neg(a);
        s = mag;
    }
    ge_proj r = ge_proj_repeat(a, s.v);
    ex_store_point(r, i, out_proj, out_aff);
}

// ---- tree reduce (one level) ------------------------------------------------------------------
// xs[odd:] = [f(xs[i], xs[i+1]) for i in range(odd, len, 2)] with odd = len % 2
__global__ void __launch_bounds__(EX_BLOCK)
k_tree_level(const uint32_t *__restrict__ in, size_t len, uint32_t *__restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t odd = len & 1;
    size_t out_len = odd + (len - odd) / 2;
    if (t >= out_len) return;
    ge_proj r;
    if (odd && t == 0) {
        r = ex_load_point(in, 0, false);
        ex_fe_st(out, r.X);
        ex_fe_st(out + 8, r.Y);
        ex_fe_st(out + 16, r.Z);
        return;
    }
    size_t i = odd + 2 * (t - odd);
    r = ge_proj_add(ex_load_point(in, i, false), ex_load_point(in, i + 1, false));
    ex_fe_st(out + 24 * t, r.X);
    ex_fe_st(out + 24 * t + 8, r.Y);
    ex_fe_st(out + 24 * t + 16, r.Z);
}

// ---- normalize / lift -------------------------------------------------------------------------
__global__ void __launch_bounds__(EX_BLOCK)
k_normalize(const uint32_t *__restrict__ proj, size_t n, uint32_t *__restrict__ out_aff) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ge_aff a = ge_proj_to_affine(ex_load_point(proj, i, false));
    ex_fe_st(out_aff + 16 * i, a.x);
    ex_fe_st(out_aff + 16 * i + 8, a.y);
}

// Montgomery's trick: one field inversion per NORM_BATCH elements instead of one each (265 of the ~275
// multiplications of a normalisation are the inversion).  A lane owns the elements t, t + lanes, t + 2 lanes, ...
// (coalesced across the wave); the running products Z_0 ... Z_k wait in the elements' own 64-byte output
// slots, so no scratch memory is needed.  An element with Z = 0 (not a point) is left out of the product and
// comes out as (0, 0), which no validation accepts.
#define NORM_BATCH 8
__global__ void __launch_bounds__(EX_BLOCK)
k_normalize_batched(const uint32_t *__restrict__ proj, size_t n, size_t lanes, uint32_t *__restrict__ out_aff) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    fe run = fe_one();
#pragma unroll 1
    for (int k = 0; k < NORM_BATCH; k++) {
        const size_t e = t + (size_t)k * lanes;
        if (e >= n) break;
        const fe Z = ex_fe_ld(proj + 24 * e + 16);
        if (!fe_is_zero(Z)) run = fe_mul(run, Z);
        ex_fe_st(out_aff + 16 * e, run);
    }
    fe inv = fe_inv(run);
#pragma unroll 1
    for (int k = NORM_BATCH - 1; k >= 0; k--) {
        const size_t e = t + (size_t)k * lanes;
        if (e >= n) continue;
        const fe Z = ex_fe_ld(proj + 24 * e + 16);
        if (fe_is_zero(Z)) {
            ex_fe_st(out_aff + 16 * e, fe_zero());
            ex_fe_st(out_aff + 16 * e + 8, fe_zero());
            continue;
        }
        const fe prev = k > 0 ? ex_fe_ld(out_aff + 16 * (e - lanes)) : fe_one();
        const fe zi = fe_mul(inv, prev);          // 1 / Z_e
        inv = fe_mul(inv, Z);                     // 1 / (Z_0 ... Z_{e-1})
        const fe x = fe_mul(ex_fe_ld(proj + 24 * e), zi), y = fe_mul(ex_fe_ld(proj + 24 * e + 8), zi);
        ex_fe_st(out_aff + 16 * e, x);
        ex_fe_st(out_aff + 16 * e + 8, y);
    }
}

// the same over a buffer of n x 96-byte points, entry point for other translation units (msm.hip)
int vmpc_normalize_launch(vmpc_ctx *ctx, const void *proj, size_t n, void *out_affine) {
    if (n == 0) return VMPC_OK;
    if (n < 4096) {        // short vectors: the chain of one inversion is the whole latency either way
        k_normalize<<<ex_grid(n), EX_BLOCK, 0, ctx->stream>>>((const uint32_t *)proj, n, (uint32_t *)out_affine);
    } else {
        const size_t lanes = (n + NORM_BATCH - 1) / NORM_BATCH;
        k_normalize_batched<<<ex_grid(lanes), EX_BLOCK, 0, ctx->stream>>>((const uint32_t *)proj, n, lanes,
                                                                         (uint32_t *)out_affine);
    }
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

__global__ void __launch_bounds__(EX_BLOCK)
k_affine_to_proj(const uint32_t *__restrict__ aff, size_t n, uint32_t *__restrict__ out_proj) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ge_proj p = ex_load_point(aff, i, true);
    ex_fe_st(out_proj + 24 * i, p.X);
    ex_fe_st(out_proj + 24 * i + 8, p.Y);
    ex_fe_st(out_proj + 24 * i + 16, p.Z);
}


extern "C" int vmpc_fold_dev(vmpc_ctx *ctx, const void *g_l, const void *g_r, int in_affine,
                             const uint8_t c[32], size_t half, void *out_proj, void *out_affine) {
    if (!ctx || !c || (half && (!g_l || !g_r)) || (!out_proj && !out_affine)) return VMPC_E_INVAL;
    if (half == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    u256_arg ca;
    memcpy(ca.v, c, 32);
    if (fr_geq_l(ca.v)) return VMPC_E_NONCANON;
    vmpc_stage_scope s(ctx, "fold");
    static const size_t quad_max = [] {
        const char *e = getenv("VMPC_FOLD_QUAD_MAX");
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)EX_FOLD_QUAD_MAX;
    }();
    static const size_t pipe_max = [] {
        const char *e = getenv("VMPC_FOLD_PIPE_MAX");
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)FP_MAX_DEFAULT;
    }();
    if (half <= pipe_max && half > (size_t)FP_ELEMS * ctx->cu_count)
        k_fold_pipe<2><<<(unsigned)((half + 2 * FP_ELEMS - 1) / (2 * FP_ELEMS)), 256, 0, ctx->stream>>>(
            (const uint32_t *)g_l, (const uint32_t *)g_r, in_affine, ca, half, (uint32_t *)out_proj,
            (uint32_t *)out_affine);
    else if (half <= pipe_max)
        k_fold_pipe<1><<<(unsigned)((half + FP_ELEMS - 1) / FP_ELEMS), 128, 0, ctx->stream>>>(
            (const uint32_t *)g_l, (const uint32_t *)g_r, in_affine, ca, half, (uint32_t *)out_proj,
            (uint32_t *)out_affine);
    else if (half <= quad_max)
        k_fold_quad<<<ex_grid(4 * half), EX_BLOCK, 0, ctx->stream>>>(
            (const uint32_t *)g_l, (const uint32_t *)g_r, in_affine, ca, half, (uint32_t *)out_proj,
            (uint32_t *)out_affine);
    else
        k_fold<<<ex_grid(half), EX_BLOCK, 0, ctx->stream>>>((const uint32_t *)g_l, (const uint32_t *)g_r,
                                                            in_affine, ca, half, (uint32_t *)out_proj,
                                                            (uint32_t *)out_affine);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_repeat_dev(vmpc_ctx *ctx, const void *bases, size_t n_bases, int bases_affine,
                               const void *scalars, size_t n, int signed_scalars, void *out_proj,
                               void *out_affine) {
    if (!ctx || (n && (!bases || !scalars)) || (!out_proj && !out_affine)) return VMPC_E_INVAL;
    if (n_bases != 1 && n_bases != n) return VMPC_E_INVAL;
    if (n == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "repeat");
    k_repeat<<<ex_grid(n), EX_BLOCK, 0, ctx->stream>>>((const uint32_t *)bases, n_bases, bases_affine,
                                                       (const uint32_t *)scalars, n, signed_scalars,
                                                       (uint32_t *)out_proj, (uint32_t *)out_affine);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_tree_reduce_dev(vmpc_ctx *ctx, void *proj_points, size_t n, int append_identity,
                                    void *out_proj) {
    if (!ctx || !out_proj || (n && !proj_points)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    size_t len = n + (append_identity ? 1 : 0);
    if (len == 0) return VMPC_E_INVAL;  // reduce() of empty sequence with no initial value
    VMPC_CHECK(vmpc_ws_reserve(ctx, 2 * vmpc_align(len * 96)));
    uint32_t *a = (uint32_t *)vmpc_ws_take(ctx, len * 96);
    uint32_t *b = (uint32_t *)vmpc_ws_take(ctx, len * 96);
    if (n) VMPC_HIP_CHECK(hipMemcpyAsync(a, proj_points, n * 96, hipMemcpyDeviceToDevice, st));
    if (append_identity) {
        uint32_t id[24] = {0};
        id[8] = 1;
        id[16] = 1;
        VMPC_HIP_CHECK(hipMemcpyAsync(a + 24 * n, id, 96, hipMemcpyHostToDevice, st));
        VMPC_HIP_CHECK(hipStreamSynchronize(st));  // `id` is a stack buffer
    }
    vmpc_stage_scope s(ctx, "tree_reduce");
    while (len > 1) {
        size_t odd = len & 1;
        size_t out_len = odd + (len - odd) / 2;
        k_tree_level<<<ex_grid(out_len), EX_BLOCK, 0, st>>>(a, len, b);
        VMPC_KERNEL_CHECK();
        uint32_t *t = a;
        a = b;
        b = t;
        len = out_len;
    }
    VMPC_HIP_CHECK(hipMemcpyAsync(out_proj, a, 96, hipMemcpyDeviceToDevice, st));
    return VMPC_OK;
}

extern "C" int vmpc_normalize_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *out_affine) {
    if (!ctx || (n && (!proj || !out_affine))) return VMPC_E_INVAL;
    if (n == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "normalize");
    return vmpc_normalize_launch(ctx, proj, n, out_affine);
}

extern "C" int vmpc_affine_to_proj_dev(vmpc_ctx *ctx, const void *affine, size_t n, void *out_proj) {
    if (!ctx || (n && (!affine || !out_proj))) return VMPC_E_INVAL;
    if (n == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    k_affine_to_proj<<<ex_grid(n), EX_BLOCK, 0, ctx->stream>>>((const uint32_t *)affine, n,
                                                               (uint32_t *)out_proj);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}
