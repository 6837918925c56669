// BN-256 G1 / G2 multi-scalar multiplication for gfx950 (BASELINE config 5).
//
// Replaces the eight sums of the reference's Pinocchio prover,
//     [int(c[i]) * evalkey[...] for i in qap.indices_mid]  +  apply_to_list(point_add, ...)
//     verifiable_mpc/trinocchio/pynocchio.py:229-246
// (one 256-bit double-and-add per term through MPyC's Jacobian arithmetic, then a tree of
// additions), by the same windowed bucket method as the Ed25519 commitment.  The scalar
// recoding, bucket sort and segment planning are shared with msm.hip (msm_sort.h); the kernels
// below are the curve-dependent half, templated on the coordinate field (F_p for G1,
// F_p[i]/(i^2+1) for the twist) and written with short-Weierstrass Jacobian formulas with
// explicit handling of the exceptional cases (P+P, P-P, infinity), which - unlike the complete
// Edwards law - can occur in bucket sums.
// Entries are affine points in Montgomery form (64 B / 128 B), accumulators Jacobian (96 B / 192 B).
#include "common.h"
#include "msm_sort.h"
#include "sw256.h"
#include "bn256_curve.h"

#define BN_B3_MONT                                                                             \
    { 0x29d50ffdu, 0x8630a1e2u, 0x5c7373e9u, 0x583653eau, 0x1867b356u, 0xabd06066u, 0x8ace581fu,  \
      0x3176f68fu }
#define BN_B2A_MONT                                                                            \
    { 0xb4c5ee14u, 0xb94f760fu, 0x4c3b6eb4u, 0xdae9f8f2u, 0xe52f4fe4u, 0x77a675d2u, 0x9116c66bu,  \
      0x736f31b0u }
#define BN_B2B_MONT                                                                            \
    { 0x386b8d71u, 0x75046774u, 0x46d36cf8u, 0x5bd0854au, 0xd41c8414u, 0x664327a1u, 0x932eeb2fu,  \
      0x096c9abbu }
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
          const uint2 *__restrict__ tasks, const uint32_t *__restrict__ n_tasks, int nb1, int seg,
          uint32_t *__restrict__ buckets, uint32_t *__restrict__ partial) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *n_tasks) return;
    uint2 tk = tasks[t];
    uint32_t ci = tk.x, sidx = tk.y;
    uint32_t cnt = counts[ci];
    uint32_t lo = starts[ci] + sidx * seg;
    uint32_t len = cnt - sidx * seg;
    if (len > (uint32_t)seg) len = seg;
    typename C::acc_t acc = C::identity();
    for (uint32_t j = 0; j < len; j++) {
        uint32_t e = sorted[lo + j];
        typename C::entry_t q = C::entry_ld(entries + (size_t)C::ENTRY_WORDS * (e & 0x7fffffffu));
        acc = C::madd(acc, q, (e >> 31) != 0);
    }
    if (nseg[ci] == 1)
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
template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
gk_reduce(const uint32_t *__restrict__ buckets, const uint32_t *__restrict__ counts, int nb, int chunks,
          int chunk_len, int log2_chunk_len, int red_blocks, uint32_t *__restrict__ partials) {
    __shared__ uint32_t lds[MSM_BLOCK * C::ACC_WORDS];
    const int w = blockIdx.y;
    const int chunk = blockIdx.x * blockDim.x + threadIdx.x;
    typename C::acc_t contrib = jac_identity<F>();
    if (chunk < chunks) {
        const int lo = chunk * chunk_len;
        const uint32_t *bw = buckets + (size_t)C::ACC_WORDS * ((size_t)w * nb + lo);
        const uint32_t *cw = counts + (size_t)w * (nb + 1) + lo + 1;
        typename C::acc_t acc = jac_identity<F>(), sum = jac_identity<F>();
        for (int j = chunk_len - 1; j >= 0; j--) {
            if (cw[j]) acc = jac_add<F>(acc, C::acc_ld(bw + (size_t)C::ACC_WORDS * j));
            sum = jac_add<F>(sum, acc);
        }
        if (chunk != 0) {
            typename C::acc_t base = acc;
            for (int k = 0; k < log2_chunk_len; k++) base = jac_dbl<F>(base);
            typename C::acc_t r = jac_identity<F>();
            int top = 31 - __clz(chunk);
            for (int k = top; k >= 0; k--) {
                r = jac_dbl<F>(r);
                if ((chunk >> k) & 1) r = jac_add<F>(r, base);
            }
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

// bucket accumulation -> finish -> reduce -> recombination; `entries` = the call's prepared points or
// a fixed-base table
template <class C, class F>
static int bn_accumulate(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, const uint32_t *entries, void *out_affine,
                         void *out_jac = nullptr) {
    hipStream_t st = ctx->stream;
    {
        vmpc_stage_scope s(ctx, "bn_bucket");
        gk_bucket<C><<<(unsigned)((w.t_max + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, st>>>(
            entries, w.sorted, w.starts, w.counts, w.nseg, w.seg_starts, w.tasks, w.ctrl + 1, p.nb1,
            (int)msm_seg_len(p), w.buckets, w.seg_partial);
        VMPC_KERNEL_CHECK();
        gk_finish_light<C, F><<<2 * ctx->cu_count, MSM_BLOCK, 0, st>>>(w.heavy_list, w.ctrl, w.nseg,
                                                                      w.seg_starts, w.seg_partial, p.nb1,
                                                                      w.buckets);
        VMPC_KERNEL_CHECK();
        gk_finish<C, F><<<2 * ctx->cu_count, MSM_BLOCK, 0, st>>>(w.heavy_list, w.ctrl, w.nseg, w.seg_starts,
                                                                w.seg_partial, p.nb1, w.buckets);
        VMPC_KERNEL_CHECK();
    }
    {
        vmpc_stage_scope s(ctx, "bn_reduce");
        gk_reduce<C, F><<<dim3(p.red_blocks, p.W), MSM_BLOCK, 0, st>>>(
            w.buckets, w.counts, p.nb, p.chunks, p.chunk_len, msm_ilog2(p.chunk_len), p.red_blocks,
            w.partials);
        VMPC_KERNEL_CHECK();
    }
    {
        vmpc_stage_scope s(ctx, "bn_final");
        gk_final<C, F><<<1, 64, 0, st>>>(w.partials, p.W, p.red_blocks, p.c, (uint32_t *)out_affine,
                                         (uint32_t *)out_jac);
        VMPC_KERNEL_CHECK();
    }
    return VMPC_OK;
}

template <class C, class F>
static int bn_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *points, size_t n, void *out_affine,
                      const char *tag) {
    if (!ctx || !out_affine || (n && (!scalars || !points))) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (n == 0) {   // empty sum = point at infinity (all-zero encoding)
        VMPC_HIP_CHECK(hipMemsetAsync(out_affine, 0, C::AFF_WORDS * 4, st));
        return VMPC_OK;
    }
    if (n >= (1ull << 31) / 17) return VMPC_E_INVAL;
    msm_plan p;
    msm_make_plan(ctx, n, 0, 256, p, &BN_ORDER);
    msm_ws w;
    msm_layout(p, w, nullptr, C::ENTRY_WORDS * 4, C::ACC_WORDS * 4);
    VMPC_CHECK(vmpc_ws_reserve(ctx, w.total));
    msm_layout(p, w, (char *)ctx->ws, C::ENTRY_WORDS * 4, C::ACC_WORDS * 4);
    {
        vmpc_stage_scope s(ctx, "bn_prep");
        gk_prep<C, F><<<(unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, st>>>(
            (const uint32_t *)points, n, w.entries);
        VMPC_KERNEL_CHECK();
    }
    VMPC_CHECK(msm_sort_stage(ctx, p, w, scalars, n, nullptr, BN_ORDER));
    (void)tag;
    return bn_accumulate<C, F>(ctx, p, w, w.entries, out_affine);
}

extern "C" int vmpc_bn256_g1_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *points, size_t n,
                                     void *out_affine) {
    return bn_msm_dev<G1, Fp1Ops>(ctx, scalars, points, n, out_affine, "g1");
}

extern "C" int vmpc_bn256_g2_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *points, size_t n,
                                     void *out_affine) {
    return bn_msm_dev<G2, Fp2Ops>(ctx, scalars, points, n, out_affine, "g2");
}

// ---- fixed-base tables (the evaluation key of a circuit is fixed: pynocchio.py:228-246 reads the
// same evalkey vectors for every proof) ---------------------------------------------------------------
// T[w][i] = 2^(16 w) * P_i for w = 0..16 as Montgomery-form affine entries; a later MSM sorts the
// flattened digit array [w][i] as ONE window whose indices are table positions, so neither the
// 256-doubling recombination (single lane: 2.6 ms for G1, 6.5 ms for the twist) nor 16/17 of the
// bucket reduction remain.  Same scheme as the Ed25519 tables in msm.hip.
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

template <class C, class F>
static int bn_table_build_dev(vmpc_ctx *ctx, const void *points, size_t n, void *table) {
    if (!ctx || !points || !table || n == 0 || n > ((size_t)1 << 26)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t stride = bn_table_stride(n);
    vmpc_stage_scope s(ctx, "bn_table_build");
    gk_table_build<C, F><<<(unsigned)((stride + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
        (const uint32_t *)points, n, stride, (uint32_t *)table);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

template <class C, class F>
static int bn_table_msm_dev(vmpc_ctx *ctx, const void *table, size_t table_n, const void *scalars, size_t m,
                            void *out_affine, void *out_jac) {
    if (!ctx || !table || (!out_affine && !out_jac) || table_n == 0 || table_n > ((size_t)1 << 26) || m > table_n ||
        (m && !scalars))
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t stride = bn_table_stride(table_n);
    msm_plan p;
    p.n_main = p.n_total = (size_t)BN_TABLE_W * stride;
    p.n_extra = 0;
    p.scalar_bits = 256;
    p.c = BN_TABLE_C;
    p.W = 1;
    p.top_row = -1;
    p.top_max_b = 0;
    p.period = 0;
    msm_plan_geometry(ctx, p);
    msm_ws w;
    msm_layout(p, w, nullptr, 0, C::ACC_WORDS * 4);
    VMPC_CHECK(vmpc_ws_reserve(ctx, w.total));
    msm_layout(p, w, (char *)ctx->ws, 0, C::ACC_WORDS * 4);
    VMPC_CHECK(msm_recode_rows(ctx, scalars, m, nullptr, 0, 0, stride, w.digits, BN_TABLE_C, BN_TABLE_W, BN_TABLE_W,
                               BN_ORDER));
    VMPC_CHECK(msm_sort_digits(ctx, p, w));
    return bn_accumulate<C, F>(ctx, p, w, (const uint32_t *)table, out_affine, out_jac);
}

extern "C" int vmpc_bn256_table_bytes(int group, size_t n, size_t *bytes) {
    if (!bytes || (group != 1 && group != 2) || n == 0 || n > ((size_t)1 << 26)) return VMPC_E_INVAL;
    *bytes = (size_t)BN_TABLE_W * bn_table_stride(n) * (group == 1 ? G1::ENTRY_WORDS : G2::ENTRY_WORDS) * 4;
    return VMPC_OK;
}

extern "C" int vmpc_bn256_table_build_dev(vmpc_ctx *ctx, int group, const void *points, size_t n, void *table) {
    if (group == 1) return bn_table_build_dev<G1, Fp1Ops>(ctx, points, n, table);
    if (group == 2) return bn_table_build_dev<G2, Fp2Ops>(ctx, points, n, table);
    return VMPC_E_INVAL;
}

extern "C" int vmpc_bn256_table_msm_dev(vmpc_ctx *ctx, int group, const void *table, size_t table_n,
                                        const void *scalars, size_t m, void *out_affine, void *out_jacobian) {
    if (group == 1) return bn_table_msm_dev<G1, Fp1Ops>(ctx, table, table_n, scalars, m, out_affine, out_jacobian);
    if (group == 2) return bn_table_msm_dev<G2, Fp2Ops>(ctx, table, table_n, scalars, m, out_affine, out_jacobian);
    return VMPC_E_INVAL;
}

extern "C" int vmpc_bn256_validate_dev(vmpc_ctx *ctx, int group, const void *points, size_t n,
                                       uint64_t *n_bad) {
    if (!ctx || !n_bad || (group != 1 && group != 2) || (n && !points)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_CHECK(vmpc_ws_reserve(ctx, 256));
    unsigned long long *d_bad = (unsigned long long *)vmpc_ws_take(ctx, 8);
    VMPC_HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    if (n) {
        unsigned g = (unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK);
        if (group == 1) {
            fp b = {BN_B3_MONT};
            gk_validate<G1, Fp1Ops><<<g, MSM_BLOCK, 0, ctx->stream>>>((const uint32_t *)points, n, b, d_bad);
        } else {
            fp2 b;
            fp ba = {BN_B2A_MONT}, bb = {BN_B2B_MONT};
            b.a = ba;
            b.b = bb;
            gk_validate<G2, Fp2Ops><<<g, MSM_BLOCK, 0, ctx->stream>>>((const uint32_t *)points, n, b, d_bad);
        }
        VMPC_KERNEL_CHECK();
    }
    unsigned long long h = 0;
    VMPC_HIP_CHECK(hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *n_bad = h;
    return VMPC_OK;
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

extern "C" int vmpc_bn256_fixed_base_dev(vmpc_ctx *ctx, int group, const void *base_affine, const void *scalars,
                                         size_t n, void *out_affine) {
    if (!ctx || (group != 1 && group != 2) || !base_affine || (n && (!scalars || !out_affine))) return VMPC_E_INVAL;
    if (n == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const unsigned g = (unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK);
    vmpc_stage_scope s(ctx, "bn_fixed_base");
    if (group == 1)
        gk_fixed_base<G1, Fp1Ops><<<g, MSM_BLOCK, 0, ctx->stream>>>((const uint32_t *)base_affine,
                                                                     (const uint32_t *)scalars, n, (uint32_t *)out_affine);
    else
        gk_fixed_base<G2, Fp2Ops><<<g, MSM_BLOCK, 0, ctx->stream>>>((const uint32_t *)base_affine,
                                                                     (const uint32_t *)scalars, n, (uint32_t *)out_affine);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

// ---- host-buffer one-shots -------------------------------------------------------------------------
static int bn_msm_host(int group, const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t *out) {
    const size_t pb = group == 1 ? 64 : 128;
    if (!out || (n && (!scalars || !points))) return VMPC_E_INVAL;
    vmpc_ctx *ctx = nullptr;
    VMPC_CHECK(vmpc_ctx_create(0, &ctx));
    void *ds = nullptr, *dp = nullptr, *dout = nullptr;
    int rc = vmpc_malloc(ctx, n * 32, &ds);
    if (!rc) rc = vmpc_malloc(ctx, n * pb, &dp);
    if (!rc) rc = vmpc_malloc(ctx, pb, &dout);
    if (!rc) rc = vmpc_memcpy_h2d(ctx, ds, scalars, n * 32);
    if (!rc) rc = vmpc_memcpy_h2d(ctx, dp, points, n * pb);
    uint64_t bad = 0;
    if (!rc) rc = vmpc_bn256_validate_dev(ctx, group, dp, n, &bad);
    if (!rc && bad) rc = VMPC_E_NOTONCURVE;
    if (!rc)
        rc = group == 1 ? vmpc_bn256_g1_msm_dev(ctx, ds, dp, n, dout) : vmpc_bn256_g2_msm_dev(ctx, ds, dp, n, dout);
    if (!rc) rc = vmpc_ctx_sync(ctx);
    if (!rc) rc = vmpc_memcpy_d2h(ctx, out, dout, pb);
    vmpc_free(ctx, ds);
    vmpc_free(ctx, dp);
    vmpc_free(ctx, dout);
    vmpc_ctx_destroy(ctx);
    return rc;
}

extern "C" int vmpc_bn256_g1_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[64]) {
    return bn_msm_host(1, scalars, points, n, out);
}
extern "C" int vmpc_bn256_g2_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[128]) {
    return bn_msm_host(2, scalars, points, n, out);
}
