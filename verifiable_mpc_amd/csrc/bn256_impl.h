// Kernels of the BN-256 G1 / G2 multi-scalar multiplication (csrc/bn256.hip holds the C-ABI and the host
// pipeline) and the table of their launchers, bn_kernels<C, F>.  The kernels over F_p^2 are large (the bucket
// kernel uses 384 VGPRs, the reduction 512 and spills) and one translation unit with all of them took ten minutes
// to compile, so the launchers are instantiated explicitly, a few per translation unit (bn256_g1.hip,
// bn256_g2_*.hip), and only DECLARED everywhere else (extern template below).
#pragma once
#include "common.h"
#include "msm_sort.h"
#include "sw256.h"
#include "bn256_curve.h"

// twist constant b' = 3 / xi, xi = i + 3 (oracle/bn256_ref.py; verifiable_mpc/ac20/pairing.py:44-51): canonical
// residues of its two coordinates, little-endian words
#define BN_B2A_CANON                                                                           \
    { 0xdb6c6949u, 0x7774124bu, 0x96e598bbu, 0x5a0cdfc5u, 0x111033b1u, 0x90e7f281u, 0x1aa5abfbu,  \
      0x64984e1fu }
#define BN_B2B_CANON                                                                           \
    { 0xd6340f0au, 0x35a2de0au, 0x83455ef6u, 0x316f8daeu, 0x7026e2d0u, 0x5dd7fe12u, 0xbaa9f3ffu,  \
      0x0e5ee696u }
static const msm_modulus BN_ORDER = {{0x57ac7261u, 0x1a2ef45bu, 0xf82b3924u, 0x2e8d8e12u, 0x6184dc21u,
                                      0xaa6fecb8u, 0x4aa387f9u, 0x8fb501e3u}};

// ---- prep: canonical affine bytes -> Montgomery-form entries ----------------------------------
template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_prep(const uint32_t *__restrict__ pts, size_t n_total, uint32_t *__restrict__ entries) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    aff<F> a = aff_load<F>(pts + (size_t)C::AFF_WORDS * i);
    C::entry_st(entries + (size_t)C::ENTRY_WORDS * i, a);
}

// ---- bucket accumulation: one lane per segment (task table from msm_sort_stage) -------------------
template <class C>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_bucket(const uint32_t *__restrict__ entries, const uint32_t *__restrict__ sorted,
          const uint32_t *__restrict__ starts, const uint32_t *__restrict__ counts,
          const uint32_t *__restrict__ nseg, const uint32_t *__restrict__ seg_starts,
          const uint2 *__restrict__ tasks, const uint32_t *__restrict__ n_tasks, int nb1, int seg, int balanced,
          uint32_t *__restrict__ buckets, uint32_t *__restrict__ partial) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_tasks) return;
    uint2 tk = tasks[t];
    uint32_t ci = tk.x, sidx = tk.y;
    const uint32_t ns = nseg[ci];
    uint32_t lo, len;
    msm_seg_range(counts[ci], ns, sidx, (uint32_t)seg, balanced, lo, len);      // msm_sort.h
    lo += starts[ci];
    typename C::acc_t acc = C::identity();
    for (uint32_t j = 0; j < len; j++) {
        uint32_t e = sorted[lo + j];
        typename C::entry_t q = C::entry_ld(entries + (size_t)C::ENTRY_WORDS * (e & 0x7fffffffu));
        acc = C::madd(acc, q, (e >> 31) != 0);
    }
    if (ns == 1)
        C::acc_st(buckets + (size_t)C::ACC_WORDS * msm_bucket_slot(ci, nb1), acc);
    else
        C::acc_st(partial + (size_t)C::ACC_WORDS * (seg_starts[ci] + sidx), acc);
}

template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_finish_light(const uint32_t *__restrict__ heavy_list, const uint32_t *__restrict__ ctrl,
                const uint32_t *__restrict__ nseg, const uint32_t *__restrict__ seg_starts,
                const uint32_t *__restrict__ partial, int nb1, uint32_t *__restrict__ buckets) {
    const uint32_t n_heavy = ctrl[0];
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n_heavy; h += gridDim.x * blockDim.x) {
        uint32_t ci = heavy_list[h];
        uint32_t ns = nseg[ci];
        if (ns > MSM_FINISH_SERIAL) continue;
        const uint32_t *src = partial + (size_t)C::ACC_WORDS * seg_starts[ci];
        typename C::acc_t acc = C::acc_ld(src);
        for (uint32_t j = 1; j < ns; j++) acc = jac_add<F>(acc, C::acc_ld(src + (size_t)C::ACC_WORDS * j));
        C::acc_st(buckets + (size_t)C::ACC_WORDS * msm_bucket_slot(ci, nb1), acc);
    }
}

template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_finish(const uint32_t *__restrict__ heavy_list, const uint32_t *__restrict__ ctrl,
          const uint32_t *__restrict__ nseg, const uint32_t *__restrict__ seg_starts,
          const uint32_t *__restrict__ partial, int nb1, uint32_t *__restrict__ buckets) {
    __shared__ uint32_t lds[MSM_BLOCK * C::ACC_WORDS];
    if (ctrl[4] == 0) return;                      // no heavily split bucket: nothing for the workgroup trees
    const uint32_t n_heavy = ctrl[0];
    for (uint32_t h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        uint32_t ci = heavy_list[h];
        uint32_t ns = nseg[ci];
        if (ns <= MSM_FINISH_SERIAL) continue;
        const uint32_t *src = partial + (size_t)C::ACC_WORDS * seg_starts[ci];
        typename C::acc_t acc = jac_identity<F>();
        for (uint32_t j = threadIdx.x; j < ns; j += blockDim.x)
            acc = jac_add<F>(acc, C::acc_ld(src + (size_t)C::ACC_WORDS * j));
        C::acc_st(lds + C::ACC_WORDS * threadIdx.x, acc);
        __syncthreads();
        for (uint32_t stride = MSM_BLOCK / 2; stride >= 1; stride >>= 1) {
            if (threadIdx.x < stride)
                C::acc_st(lds + C::ACC_WORDS * threadIdx.x,
                          jac_add<F>(C::acc_ld(lds + C::ACC_WORDS * threadIdx.x),
                                     C::acc_ld(lds + C::ACC_WORDS * (threadIdx.x + stride))));
            __syncthreads();
        }
        if (threadIdx.x == 0)
            C::acc_st(buckets + (size_t)C::ACC_WORDS * msm_bucket_slot(ci, nb1), C::acc_ld(lds));
        __syncthreads();
    }
}

// ---- reduce: sum_b b * B_b per window ---------------------------------------------------------
// SPLIT = 2 (few chunks: a 17-row table's single bucket set is 2^15 one-bucket chunks, 512 waves for 1024 SIMDs):
// TWO lanes per chunk share the offset ladder - lane kind 0 takes the low `hb` bits of the chunk index, kind 1 the
// high bits followed by hb doublings - and the workgroup tree adds the two halves like any two lanes' sums.  The
// kinds sit on whole waves (threads 0..127 / 128..255), so no wave runs both codes.  Depth 9 x (doubling + addition)
// instead of 15 x for 2^15 chunks; the workgroup covers 128 chunks, so a window leaves twice the partial sums.
template <class C, class F, int SPLIT>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_reduce(const uint32_t *__restrict__ buckets, const uint32_t *__restrict__ counts, int nb, int chunks,
          int chunk_len, int log2_chunk_len, int red_blocks, int hb, uint32_t *__restrict__ partials) {
    __shared__ uint32_t lds[MSM_BLOCK * C::ACC_WORDS];
    constexpr int PER = MSM_BLOCK / SPLIT;                 // chunks per workgroup
    const int w = blockIdx.y;
    const int kind = SPLIT == 1 ? 0 : (int)threadIdx.x / PER;
    const int chunk = blockIdx.x * PER + (int)threadIdx.x % PER;
    typename C::acc_t contrib = jac_identity<F>();
    if (chunk < chunks) {
        const int lo = chunk * chunk_len;
        const uint32_t *bw = buckets + (size_t)C::ACC_WORDS * ((size_t)w * nb + lo);
        const uint32_t *cw = counts + (size_t)w * (nb + 1) + lo + 1;
        typename C::acc_t acc = jac_identity<F>(), sum = jac_identity<F>();
        for (int j = chunk_len - 1; j >= 0; j--) {
            if (cw[j]) acc = jac_add<F>(acc, C::acc_ld(bw + (size_t)C::ACC_WORDS * j));
            if (kind == 0) sum = jac_add<F>(sum, acc);
        }
        // this lane's share of the multiplier `chunk` of base = chunk_len * acc
        const int mult = SPLIT == 1 ? chunk : (kind == 0 ? (chunk & ((1 << hb) - 1)) : (chunk >> hb));
        if (mult != 0) {
            typename C::acc_t base = acc;
            for (int k = 0; k < log2_chunk_len; k++) base = jac_dbl<F>(base);
            typename C::acc_t r = jac_identity<F>();
            int top = 31 - __clz(mult);
            for (int k = top; k >= 0; k--) {
                r = jac_dbl<F>(r);
                if ((mult >> k) & 1) r = jac_add<F>(r, base);
            }
            if (SPLIT > 1 && kind == 1)
                for (int k = 0; k < hb; k++) r = jac_dbl<F>(r);
            sum = jac_add<F>(sum, r);
        }
        contrib = sum;
    }
    C::acc_st(lds + C::ACC_WORDS * threadIdx.x, contrib);
    __syncthreads();
    for (int stride = MSM_BLOCK / 2; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride)
            C::acc_st(lds + C::ACC_WORDS * threadIdx.x,
                      jac_add<F>(C::acc_ld(lds + C::ACC_WORDS * threadIdx.x),
                                 C::acc_ld(lds + C::ACC_WORDS * (threadIdx.x + stride))));
        __syncthreads();
    }
    if (threadIdx.x == 0)
        C::acc_st(partials + (size_t)C::ACC_WORDS * ((size_t)w * red_blocks + blockIdx.x), C::acc_ld(lds));
}

// ---- final: window sums, Horner over windows, normalise ----------------------------------------
// out_jac != NULL: leave the sum in Jacobian coordinates (canonical residues X || Y || Z) - the caller
// normalises with one host inversion instead of a ~450-multiplication chain on one lane.
template <class C, class F>
__global__ void __launch_bounds__(64)
gk_final(const uint32_t *__restrict__ partials, int W, int red_blocks, int c,
         uint32_t *__restrict__ out_aff, uint32_t *__restrict__ out_jac) {
    __shared__ uint32_t lds[64 * C::ACC_WORDS];
    // block k = sum k of a multi-key pass (bn_table_msm_multi_dev): its own W * red_blocks partial sums and output slot
    partials += (size_t)C::ACC_WORDS * blockIdx.x * W * red_blocks;
    if (out_aff) out_aff += (size_t)C::AFF_WORDS * blockIdx.x;
    if (out_jac) out_jac += (size_t)3 * F::WORDS * blockIdx.x;
    // window sums: lpw lanes share a window (strided partial sums), then a short LDS tree
    int lpw = 1;
    while (lpw * 2 * W <= 64 && lpw * 2 <= red_blocks) lpw *= 2;
    const int w = threadIdx.x / lpw, sub = threadIdx.x % lpw;
    {
        typename C::acc_t r = jac_identity<F>();
        if (w < W)
            for (int j = sub; j < red_blocks; j += lpw)
                r = jac_add<F>(r, C::acc_ld(partials + (size_t)C::ACC_WORDS * ((size_t)w * red_blocks + j)));
        C::acc_st(lds + C::ACC_WORDS * threadIdx.x, r);
    }
    __syncthreads();
    for (int stride = lpw / 2; stride >= 1; stride >>= 1) {
        if (w < W && sub < stride)
            C::acc_st(lds + C::ACC_WORDS * threadIdx.x,
                      jac_add<F>(C::acc_ld(lds + C::ACC_WORDS * threadIdx.x),
                                 C::acc_ld(lds + C::ACC_WORDS * (threadIdx.x + stride))));
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        typename C::acc_t acc = C::acc_ld(lds + C::ACC_WORDS * ((W - 1) * lpw));
        for (int k = W - 2; k >= 0; k--) {
            for (int j = 0; j < c; j++) acc = jac_dbl<F>(acc);
            acc = jac_add<F>(acc, C::acc_ld(lds + C::ACC_WORDS * (k * lpw)));
        }
        if (out_aff) aff_store<F>(out_aff, jac_to_affine<F>(acc));
        if (out_jac) {
            F::store(out_jac, acc.X);
            F::store(out_jac + F::WORDS, acc.Y);
            F::store(out_jac + 2 * F::WORDS, acc.Z);
        }
    }
}

// ---- validation: canonical encodings and y^2 = x^3 + b -------------------------------------------
template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_validate(const uint32_t *__restrict__ pts, size_t n, typename F::elem b,
            unsigned long long *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = pts + (size_t)C::AFF_WORDS * i;
    bool ok = F::raw_canonical(src) && F::raw_canonical(src + F::WORDS);
    aff<F> a = aff_load<F>(src);
    if (ok && !a.inf) {
        typename F::elem lhs = F::sqr(a.y);
        typename F::elem rhs = F::add(F::mul(F::sqr(a.x), a.x), b);
        ok = F::is_zero(F::sub(lhs, rhs));
    }
    if (!ok) atomicAdd(bad, 1ull);
}

#define BN_TABLE_C 16
#define BN_TABLE_W 17

static size_t bn_table_stride(size_t n) { return (n + 7) & ~(size_t)7; }

template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_table_build(const uint32_t *__restrict__ pts, size_t n, size_t stride, uint32_t *__restrict__ table) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stride) return;
    aff<F> a;
    if (i < n) {
        a = aff_load<F>(pts + (size_t)C::AFF_WORDS * i);
    } else {                 // padding columns: never referenced (zero digits); stored as infinity
        a.inf = true;
        a.x = F::zero();
        a.y = F::zero();
    }
    C::entry_st(table + (size_t)C::ENTRY_WORDS * i, a);
    jac<F> q = jac_identity<F>();
    q = jac_madd<F>(q, a);
    for (int w = 1; w < BN_TABLE_W; w++) {
        for (int k = 0; k < BN_TABLE_C; k++) q = jac_dbl<F>(q);
        C::entry_st(table + (size_t)C::ENTRY_WORDS * ((size_t)w * stride + i), jac_to_affine<F>(q));
    }
}

// ---- fixed-base batch: out_i = n_i * B ------------------------------------------------------------------
// The evaluation / verification keys of the Pinocchio prover are n fixed-base scalar multiplications of the
// two group generators (verifiable_mpc/trinocchio/pynocchio.py:101-200 `generate_evalkey`: one `int * point`
// per key element).  One lane per element, left-to-right double-and-add over the 256 scalar bits; the
// branches of the incomplete Weierstrass law are inside jac_madd.  Affine output (one inversion per lane).
template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_fixed_base(const uint32_t *__restrict__ base, const uint32_t *__restrict__ sc, size_t n,
              uint32_t *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const aff<F> b = aff_load<F>(base);
    uint32_t s[8];
    for (int k = 0; k < 8; k++) s[k] = sc[8 * i + k];
    jac<F> acc = jac_identity<F>();
    for (int bit = 255; bit >= 0; bit--) {
        acc = jac_dbl<F>(acc);
        if ((s[bit >> 5] >> (bit & 31)) & 1u) acc = jac_madd<F>(acc, b);
    }
    aff_store<F>(out + (size_t)C::AFF_WORDS * i, jac_to_affine<F>(acc));
}


// ---- launchers ------------------------------------------------------------------------------------------------------
// two lanes per chunk when the plan's chunk-lanes would leave half of the chip's SIMDs without a wave
static inline bool bn_reduce_split(const msm_plan &p) {
    return (size_t)p.chunks * p.W <= 32768 && p.chunks >= 512 && (p.chunks & (p.chunks - 1)) == 0 &&
           p.chunks % (MSM_BLOCK / 2) == 0;
}

template <class C, class F>
struct bn_kernels {
    static int prep(vmpc_ctx *ctx, const void *points, size_t n, uint32_t *entries);
    static int bucket(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, const uint32_t *entries);
    static int reduce(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w);
    static int final(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, void *out_affine, void *out_jac);
    // K one-window sums side by side (w.partials: K x red_blocks), Jacobian out, K x 3 field elements
    static int final_multi(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, void *out_jac, int K);
    static int table_build(vmpc_ctx *ctx, const void *points, size_t n, size_t stride, void *table);
    static int validate(vmpc_ctx *ctx, const void *points, size_t n, unsigned long long *d_bad);
    static int fixed_base(vmpc_ctx *ctx, const void *base_affine, const void *scalars, size_t n, void *out_affine);
};

template <class C, class F>
int bn_kernels<C, F>::prep(vmpc_ctx *ctx, const void *points, size_t n, uint32_t *entries) {
    gk_prep<C, F><<<(unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>((const uint32_t *)points, n,
                                                                                              entries);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

template <class C, class F>
int bn_kernels<C, F>::bucket(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, const uint32_t *entries) {
    hipStream_t st = ctx->stream;
    gk_bucket<C><<<(unsigned)((w.t_max + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, st>>>(
        entries, w.sorted, w.starts, w.counts, w.nseg, w.seg_starts, w.tasks, w.ctrl + 1, p.nb1,
        (int)msm_seg_len(p), p.balanced, w.buckets, w.seg_partial);
    VMPC_KERNEL_CHECK();
    gk_finish_light<C, F><<<2 * ctx->cu_count, MSM_BLOCK, 0, st>>>(w.heavy_list, w.ctrl, w.nseg, w.seg_starts,
                                                                  w.seg_partial, p.nb1, w.buckets);
    VMPC_KERNEL_CHECK();
    gk_finish<C, F><<<2 * ctx->cu_count, MSM_BLOCK, 0, st>>>(w.heavy_list, w.ctrl, w.nseg, w.seg_starts,
                                                            w.seg_partial, p.nb1, w.buckets);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

template <class C, class F>
int bn_kernels<C, F>::reduce(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w) {
    if (bn_reduce_split(p)) {
        // balance 23 hb (low kind: hb ladder steps) against 23 (bits - hb) + 7 hb (high kind: its steps, then hb doublings)
        const int bits = msm_ilog2(p.chunks), hb = (23 * bits + 19) / 39;
        gk_reduce<C, F, 2><<<dim3(2 * p.red_blocks, p.W), MSM_BLOCK, 0, ctx->stream>>>(
            w.buckets, w.counts, p.nb, p.chunks, p.chunk_len, msm_ilog2(p.chunk_len), 2 * p.red_blocks, hb, w.partials);
    } else {
        gk_reduce<C, F, 1><<<dim3(p.red_blocks, p.W), MSM_BLOCK, 0, ctx->stream>>>(
            w.buckets, w.counts, p.nb, p.chunks, p.chunk_len, msm_ilog2(p.chunk_len), p.red_blocks, 0, w.partials);
    }
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

template <class C, class F>
int bn_kernels<C, F>::final(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, void *out_affine, void *out_jac) {
    gk_final<C, F><<<1, 64, 0, ctx->stream>>>(w.partials, p.W, bn_reduce_split(p) ? 2 * p.red_blocks : p.red_blocks, p.c,
                                              (uint32_t *)out_affine, (uint32_t *)out_jac);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

template <class C, class F>
int bn_kernels<C, F>::final_multi(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, void *out_jac, int K) {
    gk_final<C, F><<<K, 64, 0, ctx->stream>>>(w.partials, 1, bn_reduce_split(p) ? 2 * p.red_blocks : p.red_blocks, p.c,
                                              nullptr, (uint32_t *)out_jac);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

template <class C, class F>
int bn_kernels<C, F>::table_build(vmpc_ctx *ctx, const void *points, size_t n, size_t stride, void *table) {
    gk_table_build<C, F><<<(unsigned)((stride + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
        (const uint32_t *)points, n, stride, (uint32_t *)table);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

template <class C, class F>
int bn_kernels<C, F>::fixed_base(vmpc_ctx *ctx, const void *base_affine, const void *scalars, size_t n,
                                 void *out_affine) {
    gk_fixed_base<C, F><<<(unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
        (const uint32_t *)base_affine, (const uint32_t *)scalars, n, (uint32_t *)out_affine);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

// the curve constant b (3 for G1, 3 / (i + 3) for the twist) in the field's Montgomery form, from its canonical value
inline fp29 bn_curve_b(const BnF1 *) {
    const uint32_t three[8] = {3, 0, 0, 0, 0, 0, 0, 0};
    return BnF1::load(three);
}
inline fp29x2 bn_curve_b(const BnF2 *) {
    // b' = 3 / (i + 3)
    const uint32_t ba[8] = BN_B2A_CANON, bb[8] = BN_B2B_CANON;
    fp29x2 b;
    b.a = BnF1::load(ba);
    b.b = BnF1::load(bb);
    return b;
}

template <class C, class F>
int bn_kernels<C, F>::validate(vmpc_ctx *ctx, const void *points, size_t n, unsigned long long *d_bad) {
    gk_validate<C, F><<<(unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
        (const uint32_t *)points, n, bn_curve_b((const F *)nullptr), d_bad);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

// instantiated in bn256_g1.hip / bn256_g2_*.hip
extern template struct bn_kernels<G1, BnF1>;
extern template struct bn_kernels<G2, BnF2>;
