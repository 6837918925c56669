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
// This file: the C-ABI and the host side of the pipeline; the kernels and their launchers are in bn256_impl.h and
// are compiled in bn256_g1.hip / bn256_g2_*.hip.
#include "bn256_impl.h"

// bucket accumulation -> finish -> reduce -> recombination; `entries` = the call's prepared points or
// a fixed-base table
template <class C, class F>
static int bn_accumulate(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, const uint32_t *entries, void *out_affine,
                         void *out_jac = nullptr) {
    {
        vmpc_stage_scope s(ctx, "bn_bucket");
        VMPC_CHECK((bn_kernels<C, F>::bucket(ctx, p, w, entries)));
    }
    {
        vmpc_stage_scope s(ctx, "bn_reduce");
        VMPC_CHECK((bn_kernels<C, F>::reduce(ctx, p, w)));
    }
    {
        vmpc_stage_scope s(ctx, "bn_final");
        VMPC_CHECK((bn_kernels<C, F>::final(ctx, p, w, out_affine, out_jac)));
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
        VMPC_CHECK((bn_kernels<C, F>::prep(ctx, points, n, w.entries)));
    }
    VMPC_CHECK(msm_sort_stage(ctx, p, w, scalars, n, nullptr, BN_ORDER));
    (void)tag;
    return bn_accumulate<C, F>(ctx, p, w, w.entries, out_affine);
}

extern "C" int vmpc_bn256_g1_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *points, size_t n,
                                     void *out_affine) {
    return bn_msm_dev<G1, BnF1>(ctx, scalars, points, n, out_affine, "g1");
}

extern "C" int vmpc_bn256_g2_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *points, size_t n,
                                     void *out_affine) {
    return bn_msm_dev<G2, BnF2>(ctx, scalars, points, n, out_affine, "g2");
}

// ---- fixed-base tables (the evaluation key of a circuit is fixed: pynocchio.py:228-246 reads the
// same evalkey vectors for every proof) ---------------------------------------------------------------
// T[w][i] = 2^(16 w) * P_i for w = 0..16 as Montgomery-form affine entries; a later MSM sorts the
// flattened digit array [w][i] as ONE window whose indices are table positions, so neither the
// 256-doubling recombination (single lane: 2.6 ms for G1, 6.5 ms for the twist) nor 16/17 of the
// bucket reduction remain.  Same scheme as the Ed25519 tables in msm.hip.
template <class C, class F>
static int bn_table_build_dev(vmpc_ctx *ctx, const void *points, size_t n, void *table) {
    if (!ctx || !points || !table || n == 0 || n > ((size_t)1 << 26)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t stride = bn_table_stride(n);
    vmpc_stage_scope s(ctx, "bn_table_build");
    return bn_kernels<C, F>::table_build(ctx, points, n, stride, table);
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

// ---- several prepared keys, ONE scalar vector (the six G1 sums and the twist sum of pynocchio.py:229-246 all run over
// c_mid): the digits are recoded, sorted and planned once; every table gets its own bucket launch over the SAME sorted
// index list (an index is a table position, and the tables share their geometry), its own bucket array and partial
// sums; the K bucket sets are then reduced as the K windows of one launch and finished by K workgroups side by side.
// Per sum that saves the recoding, the sort and the plan (~0.11 ms of a 1.0-ms G1 sum) and, more, the serial tails:
// one reduction and one recombination launch for all K instead of K of each, taking turns with bucket kernels.
// A column a sum does not use (the zero-knowledge terms are per element) holds the point at infinity in that table:
// the mixed addition skips it (SwCurve::entry_ld marks the all-zero entry), so the K sums may differ in which of the
// trailing columns they include while sharing every scalar.
template <class C, class F>
static int bn_table_msm_multi_dev(vmpc_ctx *ctx, const void *const *tables, int K, size_t table_n, const void *scalars,
                                  size_t m, void *out_jac) {
    if (!ctx || !tables || K < 1 || K > 16 || !out_jac || table_n == 0 || table_n > ((size_t)1 << 26) || m > table_n ||
        (m && !scalars))
        return VMPC_E_INVAL;
    for (int k = 0; k < K; k++)
        if (!tables[k]) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
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
    msm_plan pk = p;                      // the K bucket sets as K windows of the reduction
    pk.W = K;
    // its chunk-lanes: the 512-VGPR reduction keeps 2^16 lanes resident (one wave per SIMD); beyond that a launch runs
    // in rounds and every lane repeats the offset ladder (msm_plan_geometry's rule, applied to K windows)
    while ((size_t)pk.chunks * K > (size_t)MSM_REDUCE_CHUNKS * 16 && pk.chunks > 256) pk.chunks /= 2;
    pk.chunk_len = pk.nb / pk.chunks;
    pk.red_blocks = (pk.chunks + MSM_BLOCK - 1) / MSM_BLOCK;
    msm_ws w;
    msm_layout(p, w, nullptr, 0, C::ACC_WORDS * 4);
    const size_t acc = (size_t)C::ACC_WORDS * 4;
    const size_t rb = (size_t)(bn_reduce_split(pk) ? 2 * pk.red_blocks : pk.red_blocks);
    const size_t b_bytes = vmpc_align((size_t)K * p.nb * acc);
    const size_t c_bytes = vmpc_align((size_t)K * p.nb1 * 4), r_bytes = vmpc_align((size_t)K * rb * acc);
    const size_t base = vmpc_align(w.total);
    VMPC_CHECK(vmpc_ws_reserve(ctx, base + b_bytes + c_bytes + r_bytes));
    msm_layout(p, w, (char *)ctx->ws, 0, C::ACC_WORDS * 4);
    char *extra = (char *)ctx->ws + base;
    uint32_t *buckets = (uint32_t *)extra;
    uint32_t *counts = (uint32_t *)(extra + b_bytes);
    uint32_t *partials = (uint32_t *)(extra + b_bytes + c_bytes);
    VMPC_CHECK(msm_recode_rows(ctx, scalars, m, nullptr, 0, 0, stride, w.digits, BN_TABLE_C, BN_TABLE_W, BN_TABLE_W,
                               BN_ORDER));
    VMPC_CHECK(msm_sort_digits(ctx, p, w));
    {
        vmpc_stage_scope s(ctx, "bn_bucket");
        for (int k = 0; k < K; k++) {
            msm_ws wk = w;
            wk.buckets = buckets + (size_t)k * p.nb * C::ACC_WORDS;
            // (w.seg_partial is shared: a table's finish kernels have consumed it before the next bucket launch)
            VMPC_CHECK((bn_kernels<C, F>::bucket(ctx, p, wk, (const uint32_t *)tables[k])));
            // the reduction reads window k's bucket counts at counts + k * nb1: the same counts for every table
            VMPC_HIP_CHECK(hipMemcpyAsync(counts + (size_t)k * p.nb1, w.counts, (size_t)p.nb1 * 4, hipMemcpyDeviceToDevice, st));
        }
    }
    msm_ws wm = w;
    wm.buckets = buckets;
    wm.counts = counts;
    wm.partials = partials;
    {
        vmpc_stage_scope s(ctx, "bn_reduce");
        VMPC_CHECK((bn_kernels<C, F>::reduce(ctx, pk, wm)));
    }
    {
        vmpc_stage_scope s(ctx, "bn_final");
        VMPC_CHECK((bn_kernels<C, F>::final_multi(ctx, pk, wm, out_jac, K)));
    }
    return VMPC_OK;
}

extern "C" int vmpc_bn256_table_msm_multi_dev(vmpc_ctx *ctx, int group, const void *const *tables, int n_tables,
                                              size_t table_n, const void *scalars, size_t m, void *out_jacobian) {
    if (group == 1) return bn_table_msm_multi_dev<G1, BnF1>(ctx, tables, n_tables, table_n, scalars, m, out_jacobian);
    if (group == 2) return bn_table_msm_multi_dev<G2, BnF2>(ctx, tables, n_tables, table_n, scalars, m, out_jacobian);
    return VMPC_E_INVAL;
}

extern "C" int vmpc_bn256_table_bytes(int group, size_t n, size_t *bytes) {
    if (!bytes || (group != 1 && group != 2) || n == 0 || n > ((size_t)1 << 26)) return VMPC_E_INVAL;
    *bytes = (size_t)BN_TABLE_W * bn_table_stride(n) * (group == 1 ? G1::ENTRY_WORDS : G2::ENTRY_WORDS) * 4;
    return VMPC_OK;
}

extern "C" int vmpc_bn256_table_build_dev(vmpc_ctx *ctx, int group, const void *points, size_t n, void *table) {
    if (group == 1) return bn_table_build_dev<G1, BnF1>(ctx, points, n, table);
    if (group == 2) return bn_table_build_dev<G2, BnF2>(ctx, points, n, table);
    return VMPC_E_INVAL;
}

extern "C" int vmpc_bn256_table_msm_dev(vmpc_ctx *ctx, int group, const void *table, size_t table_n,
                                        const void *scalars, size_t m, void *out_affine, void *out_jacobian) {
    if (group == 1) return bn_table_msm_dev<G1, BnF1>(ctx, table, table_n, scalars, m, out_affine, out_jacobian);
    if (group == 2) return bn_table_msm_dev<G2, BnF2>(ctx, table, table_n, scalars, m, out_affine, out_jacobian);
    return VMPC_E_INVAL;
}

extern "C" int vmpc_bn256_validate_dev(vmpc_ctx *ctx, int group, const void *points, size_t n,
                                       uint64_t *n_bad) {
    if (!ctx || !n_bad || (group != 1 && group != 2) || (n && !points)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_CHECK(vmpc_ws_reserve(ctx, 256));
    unsigned long long *d_bad = (unsigned long long *)vmpc_ws_take(ctx, 8);
    VMPC_HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    if (n)
        VMPC_CHECK(group == 1 ? (bn_kernels<G1, BnF1>::validate(ctx, points, n, d_bad))
                              : (bn_kernels<G2, BnF2>::validate(ctx, points, n, d_bad)));
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
extern "C" int vmpc_bn256_fixed_base_dev(vmpc_ctx *ctx, int group, const void *base_affine, const void *scalars,
                                         size_t n, void *out_affine) {
    if (!ctx || (group != 1 && group != 2) || !base_affine || (n && (!scalars || !out_affine))) return VMPC_E_INVAL;
    if (n == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "bn_fixed_base");
    return group == 1 ? bn_kernels<G1, BnF1>::fixed_base(ctx, base_affine, scalars, n, out_affine)
                      : bn_kernels<G2, BnF2>::fixed_base(ctx, base_affine, scalars, n, out_affine);
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
