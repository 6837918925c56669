// Short commitments over a 16-row fixed-base table in THREE launches (round 5).
//
// A Pedersen commitment  h^gamma * prod g_i^{x_i}  (verifiable_mpc/ac20/pivot.py:139-145) over a tabulated CRS of at
// most 2^17 columns - config 2 of BASELINE.json (n = 2^16), and every round of the compact prover after its fold jump
// (the A_i, B_i pair of compressed_pivot.py:41-42 over a 2^15-generator table) - is latency, not throughput: the
// general pipeline (msm.hip / msm_sort.hip) spends ~17 dependent launches on it (recode, histogram, three scans,
// partition, fine sort, plan, bucket, finish, reduction tree, combine), 0.24-0.25 ms for 60 us of arithmetic.
// With 16 rows a commitment has ONE set of 2^15 buckets.  Here:
//
//   k_short_scatter   lane per table column: recodes the column's scalar (16 signed 16-bit digits, digit w belongs to
//                     row w) and appends an entry {table position, fine bucket, sign} to the global list of its BIN
//                     (128 bins of 256 buckets; list slots reserved with one LDS-aggregated atomic per bin and workgroup)
//   k_short_bins      workgroup per (bin, part): counting-sorts its entries by bucket in LDS, accumulates every bucket
//                     with two lanes (mixed additions from the table), a bucket with more than SH_HEAVY entries with
//                     the whole workgroup, and runs the quad weight tree (rt_tree.h) over its 256 bucket sums
//   k_msm_reduce_combine   (msm_reduce_tree.hip) adds the parts' triples, finishes sum_b b * B_b over the 128 bins,
//                     writes the commitment in the public 128-byte form, publishes the completion word of a queued
//                     prover round and re-arms the bin cursors
//
// One commitment is shared out among 2 x 128 workgroups (parts = entries of even / odd list position), a pair among
// 2 x 128: the chip's 256 CUs each get one.  Capacities are fixed (SH_T entries per workgroup, SH_MAX_HEAVY whole-
// workgroup buckets); an input beyond them - NOT only an adversarial one: a witness whose non-zero digits are mostly
// small values (many 0/1/2 wires) fills bin 0 alone, > ~24 K such wires in a vector of 2^14.6 .. 2^17 - raises
// VMPC_ST_SHORT_OVERFLOW, vmpc_ctx_sync answers VMPC_E_AGAIN and the caller repeats the call on the general path
// (verifiable_mpc_amd/pivot.py, compressed_pivot.py); the void result itself goes out as all zeros (Z = 0 is no point),
// so that with several commitments pending on one context each can tell whether it was the one.  The group element is the same as the general path's; its
// extended representative (X : Y : Z : T) is not defined (lists are filled in atomic order) - every consumer
// normalises.
#include "common.h"
#include "msm_sort.h"
#include "fe25519.h"
#include "ge25519.h"
#include "ptio.h"
#include "quad.h"
#include "rt_tree.h"

#define SH_BINS 128
#define SH_FINE 256
#define SH_T 12288             // entries per workgroup: 48 KB of LDS sorted + staging shared with the tree's 80 KB
#define SH_THREADS 512
#define SH_HEAVY 192
#define SH_MAX_HEAVY 8
#define SH_POS_BITS 21
#define SH_STAGE_WORDS (2 * RT_LEAVES * EXT_WORDS)      // 20480 words = 80 KB: entries as loaded | heavy partials | TA, RR, DD

struct sh_scalars {
    const uint32_t *sc[2];
    const uint32_t *sc_extra[2];
};

// ---- 1. recode + bin -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(MSM_BLOCK)
k_short_scatter(sh_scalars a, size_t m, size_t table_n, size_t n_extra, size_t stride, msm_modulus mod, uint32_t cap,
                uint32_t *__restrict__ cursors, uint32_t *__restrict__ bins, uint32_t *__restrict__ status,
                uint32_t *__restrict__ poison) {
    __shared__ uint32_t cnt[SH_BINS], base[SH_BINS];
    const int k = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) poison[k] = 0;      // this call's own overflow word (set by k_short_bins)
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < SH_BINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t *src = nullptr;
    if (col < m) src = a.sc[k] + 8 * col;
    else if (col >= table_n && col < table_n + n_extra && a.sc_extra[k]) src = a.sc_extra[k] + 8 * (col - table_n);
    // tag: sign << 31 | fine bucket << 23 | bin << 16 | rank inside this workgroup's run of the bin; ~0 = no entry
    uint32_t tag[16];
#pragma unroll
    for (int w = 0; w < 16; w++) tag[w] = 0xffffffffu;
    if (src) {
        uint32_t s[8];
        load_u32x8(s, src);
        bool ge = true;         // canonical residue?  (as msm_recode_term: the term counts as zero, the call fails at its sync)
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            if (s[i] != mod.v[i]) {
                ge = s[i] > mod.v[i];
                break;
            }
        }
        if (ge) {
            atomicAdd(&status[VMPC_ST_NONCANON], 1u);
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = 0;
        }
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t raw = ((s[w >> 1] >> (16 * (w & 1))) & 0xffffu) + carry;
            int32_t d;
            if (raw >= 0x8000u) {
                d = (int32_t)raw - 0x10000;
                carry = 1;
            } else {
                d = (int32_t)raw;
                carry = 0;
            }
            if (d != 0) {
                const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;        // bucket 0 .. 2^15 - 1
                const uint32_t bin = b >> 8;
                tag[w] = (d < 0 ? 0x80000000u : 0u) | ((b & 255u) << 23) | (bin << 16) | atomicAdd(&cnt[bin], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < SH_BINS) {
        const uint32_t c = cnt[threadIdx.x];
        base[threadIdx.x] = c ? atomicAdd(&cursors[k * SH_BINS + threadIdx.x], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t t = tag[w];
        if (t != 0xffffffffu) {
            const uint32_t bin = (t >> 16) & 127u, idx = base[bin] + (t & 0xffffu);
            if (idx < cap)       // beyond the capacity: dropped, the bin's workgroup sees cursor > cap and raises the flag
                bins[((size_t)k * SH_BINS + bin) * cap + idx] =
                    (uint32_t)((size_t)w * stride + col) | (((t >> 23) & 255u) << SH_POS_BITS) | ((t >> 31) << 29);
        }
    }
}

// ---- 2. per bin: sort by bucket, accumulate, weight tree -----------------------------------------------------------
__device__ __forceinline__ ge_niels sh_entry(const uint32_t *__restrict__ table, uint32_t e) {
    return niels_ld_line(table + NIELS_WORDS * (size_t)(e & ((1u << SH_POS_BITS) - 1u)));
}
// sum of the entries srt[first], srt[first + step], ... below `end`, the NEXT table line requested before the current
// addition starts: with two waves per SIMD the gather's latency is otherwise on the chain (6 us per addition measured,
// 2.2 us of it arithmetic)
__device__ __forceinline__ ge_ext sh_sum(const uint32_t *__restrict__ table, const uint32_t *srt, uint32_t first,
                                         uint32_t end, uint32_t step) {
    ge_ext acc = ge_ext_identity();
    if (first >= end) return acc;
    uint32_t e = srt[first];
    ge_niels q = sh_entry(table, e);
    for (uint32_t j = first; j < end; j += step) {
        const uint32_t jn = j + step < end ? j + step : j;
        const uint32_t en = srt[jn];
        const ge_niels qn = sh_entry(table, en);
        acc = ge_madd(acc, ge_niels_select_neg(q, ((e >> 29) & 1u) != 0));
        e = en;
        q = qn;
    }
    return acc;
}

__global__ void __launch_bounds__(SH_THREADS)
k_short_bins(const uint32_t *__restrict__ table, const uint32_t *__restrict__ bins, const uint32_t *__restrict__ cursors,
             uint32_t cap, int spl, uint32_t *__restrict__ out3, uint32_t *__restrict__ status,
             uint32_t *__restrict__ poison) {
    extern __shared__ __align__(16) uint32_t sh_lds[];
    uint32_t *srt = sh_lds;                        // [SH_T] entries in bucket order
    uint32_t *stage = srt + SH_T;                  // [SH_STAGE_WORDS]
    uint32_t *cnt = stage + SH_STAGE_WORDS;        // [256] entries per bucket
    uint32_t *start = cnt + SH_FINE;               // [256]
    uint32_t *cur = start + SH_FINE;               // [256]
    uint32_t *heavy = cur + SH_FINE;               // [0] = how many, [1 ..] = which buckets
    const int g = blockIdx.x, k = blockIdx.y / spl, s = blockIdx.y % spl;
    const int tid = threadIdx.x;
    const uint32_t have = cursors[k * SH_BINS + g];
    const uint32_t n = have < cap ? have : cap;
    uint32_t mine = n > (uint32_t)s ? (n - (uint32_t)s + (uint32_t)spl - 1u) / (uint32_t)spl : 0u;
    bool overflow = have > cap;
    if (mine > SH_T) {
        mine = SH_T;
        overflow = true;
    }
    if (tid < SH_FINE) cnt[tid] = 0;
    if (tid == 0) heavy[0] = 0;
    __syncthreads();
    const uint32_t *src = bins + ((size_t)k * SH_BINS + g) * cap;
    for (uint32_t i = tid; i < mine; i += SH_THREADS) {
        const uint32_t e = src[(size_t)s + (size_t)i * spl];
        stage[i] = e;
        atomicAdd(&cnt[(e >> SH_POS_BITS) & 255u], 1u);
    }
    __syncthreads();
    // exclusive scan of the 256 counts: inside each of the first four waves by shuffles, the waves' totals through LDS
    uint32_t own = 0, incl = 0;
    if (tid < SH_FINE) {
        own = cnt[tid];
        incl = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += up;
        }
        if ((tid & 63) == 63) cur[tid >> 6] = incl;          // (cur is rewritten below)
    }
    __syncthreads();
    if (tid < SH_FINE) {
        uint32_t before = 0;
        for (int w = 0; w < (tid >> 6); w++) before += cur[w];
        start[tid] = incl + before;
    }
    __syncthreads();
    if (tid < SH_FINE) {
        const uint32_t ex = start[tid] - own;
        cur[tid] = ex;
        if (own > SH_HEAVY) {
            const uint32_t slot = atomicAdd(&heavy[0], 1u);
            if (slot < SH_MAX_HEAVY) heavy[1 + slot] = (uint32_t)tid;
        }
    }
    __syncthreads();
    if (tid < SH_FINE) start[tid] = cur[tid];
    __syncthreads();
    for (uint32_t i = tid; i < mine; i += SH_THREADS) {
        const uint32_t e = stage[i];
        srt[atomicAdd(&cur[(e >> SH_POS_BITS) & 255u], 1u)] = e;
    }
    __syncthreads();
    uint32_t n_heavy = heavy[0];
    if (n_heavy > SH_MAX_HEAVY) {
        n_heavy = SH_MAX_HEAVY;
        overflow = true;
    }
    if (overflow && tid == 0) {
        atomicAdd(&status[VMPC_ST_SHORT_OVERFLOW], 1u);
        poison[k] = 1;               // the combine kernel voids THIS commitment's output
    }
    // every bucket: two lanes, entries of even / odd rank
    const int b = tid >> 1, h = tid & 1;
    ge_ext acc;
    {
        const uint32_t c = cnt[b], s0 = start[b];
        acc = sh_sum(table, srt, s0 + (uint32_t)h, c <= SH_HEAVY ? s0 + c : s0, 2);
    }
    // a bucket that holds a large share of the workgroup's entries (small witness values: most scalars of a circuit's
    // wire vector are 0, 1 or 2, circuit_sat_cb.py:91-103): a group of lanes sums it in strides, the partial sums go
    // through a quad tree in the staging area, the bucket's first lane takes the result
    // ALL of the workgroup's heavy buckets at once: 512 / (their number, rounded up to a power of two) lanes each
    if (n_heavy) {
        int groups = 1;
        while ((uint32_t)groups < n_heavy) groups <<= 1;
        const int lanes_per = SH_THREADS / groups;
        const int gi = tid / lanes_per, li = tid % lanes_per;
        ge_ext part = ge_ext_identity();
        if ((uint32_t)gi < n_heavy) {
            const uint32_t hb = heavy[1 + gi], c = cnt[hb], s0 = start[hb];
            part = sh_sum(table, srt, s0 + (uint32_t)li, s0 + c, (uint32_t)lanes_per);
        }
        ext_st(stage + EXT_WORDS * tid, part);
        __syncthreads();
        const int q = tid & 3;
        for (int mm = lanes_per / 2; mm >= 1; mm >>= 1) {
            for (int job = tid >> 2; job < groups * mm; job += SH_THREADS / 4) {
                const int base = (job / mm) * lanes_per, j = job % mm;
                rt_st(stage, base + j, q, rt_add(rt_ld(stage, base + j, q), rt_ld(stage, base + j + mm, q), q));
            }
            __syncthreads();
        }
        for (uint32_t hi = 0; hi < n_heavy; hi++)
            if (tid == 2 * (int)heavy[1 + hi]) acc = ge_add(acc, ext_ld(stage + EXT_WORDS * (size_t)hi * lanes_per));
    }
    __syncthreads();
    // the two lanes of a bucket -> its sum, as leaf b of the tree
    uint32_t *TA = stage, *RR = TA + RT_LEAVES * EXT_WORDS, *DD = RR + (RT_LEAVES / 2) * EXT_WORDS;
    {
        ge_ext other;
        other.X = quad_perm<0xB1>(acc.X);
        other.Y = quad_perm<0xB1>(acc.Y);
        other.Z = quad_perm<0xB1>(acc.Z);
        other.T = quad_perm<0xB1>(acc.T);
        const ge_ext tot = ge_add(acc, other);
        if (h == 0) ext_st(TA + EXT_WORDS * b, tot);
    }
    __syncthreads();
    rt_tree<true, false, false>(TA, nullptr, nullptr, RR, DD, RT_LEAVES);
    uint32_t *o = out3 + (size_t)EXT_WORDS * 3 * (((size_t)k * SH_BINS + g) * spl + s);
    if (tid < EXT_WORDS) {
        o[tid] = TA[tid];
        o[EXT_WORDS + tid] = RR[tid];
        o[2 * EXT_WORDS + tid] = DD[tid];
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
int msm_reduce_combine_launch(vmpc_ctx *ctx, const uint32_t *triples, int W, int G, int spl, uint32_t *scratch_out,
                              void *out_packed, uint32_t *reset, int reset_words, const uint32_t *poison);  // msm_reduce_tree.hip

bool msm_short_fits(const vmpc_ctx *ctx, size_t table_n, size_t table_extra, size_t m, int rows, int c, int K,
                    const void *out_ext, const void *out_affine) {
    const size_t stride = (table_n + table_extra + 7) & ~(size_t)7;
    // uniform scalars put 16 (m + extras) / 128 entries into a bin; a workgroup (half a bin for one commitment, a
    // whole one for a pair) must expect at most two thirds of its SH_T: one commitment up to 2^17 terms, a pair 2^16
    const size_t per_wg = 16 * (m + table_extra) / SH_BINS / (K == 1 ? 2 : 1);
    const size_t lds_bytes = ((size_t)SH_T + SH_STAGE_WORDS + 3 * SH_FINE + 16) * 4;     // k_short_bins' dynamic LDS
    return ctx->short_path && ctx->lds_optin >= lds_bytes && !ctx->bucket_stream && rows == 16 && c == 16 && K >= 1 && K <= 2 && out_ext && !out_affine &&
           (size_t)16 * stride <= ((size_t)1 << SH_POS_BITS) && per_wg <= (size_t)SH_T * 2 / 3 + 16;
}

int msm_short_batch(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, const void *const *scalars,
                    size_t m, const void *const *extra_scalars, int K, void *out_ext, const msm_modulus &modulus) {
    hipStream_t st = ctx->stream;
    const size_t stride = (table_n + table_extra + 7) & ~(size_t)7;
    const int spl = K == 1 ? 2 : 1;
    const uint32_t cap = (uint32_t)spl * SH_T;
    const size_t lds_bytes = ((size_t)SH_T + SH_STAGE_WORDS + 3 * SH_FINE + 16) * 4;
    if (!ctx->short_cursors) {
        VMPC_HIP_CHECK(hipMalloc((void **)&ctx->short_cursors, 2 * SH_BINS * sizeof(uint32_t)));
        VMPC_HIP_CHECK(hipMemsetAsync(ctx->short_cursors, 0, 2 * SH_BINS * sizeof(uint32_t), st));
    }
    if (!ctx->short_ready) {
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_short_bins, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes));
    }
    const size_t bins_bytes = vmpc_align((size_t)K * SH_BINS * cap * 4);
    const size_t tri_bytes = vmpc_align((size_t)K * SH_BINS * spl * 3 * EXT_WORDS * 4);
    const size_t out_bytes = vmpc_align((size_t)K * EXT_WORDS * 4);
    VMPC_CHECK(vmpc_ws_reserve(ctx, bins_bytes + tri_bytes + out_bytes + 2048));
    uint32_t *bins = (uint32_t *)vmpc_ws_take(ctx, bins_bytes);
    uint32_t *triples = (uint32_t *)vmpc_ws_take(ctx, tri_bytes);
    uint32_t *scratch = (uint32_t *)vmpc_ws_take(ctx, out_bytes);
    uint32_t *poison = (uint32_t *)vmpc_ws_take(ctx, 256);
    sh_scalars a;
    for (int k = 0; k < 2; k++) {
        a.sc[k] = k < K ? (const uint32_t *)scalars[k] : nullptr;
        a.sc_extra[k] = (k < K && extra_scalars) ? (const uint32_t *)extra_scalars[k] : nullptr;
    }
    const size_t n_cols = table_n + table_extra;
    {
        vmpc_stage_scope s(ctx, "short_scatter");
        k_short_scatter<<<dim3((unsigned)((n_cols + MSM_BLOCK - 1) / MSM_BLOCK), K), MSM_BLOCK, 0, st>>>(
            a, m, table_n, table_extra, stride, modulus, cap, ctx->short_cursors, bins, ctx->d_status, poison);
        VMPC_KERNEL_CHECK();
    }
    // from here on the bin cursors hold this call's counts and only the LAST kernel re-arms them: a launch that fails
    // in between must not leave them to the next call
    int rc = VMPC_OK;
    {
        vmpc_stage_scope s(ctx, "short_bins");
        k_short_bins<<<dim3(SH_BINS, K * spl), SH_THREADS, lds_bytes, st>>>((const uint32_t *)table, bins, ctx->short_cursors,
                                                                          cap, spl, triples, ctx->d_status, poison);
        if (hipGetLastError() != hipSuccess) rc = VMPC_E_HIP;
    }
    if (rc == VMPC_OK) {
        vmpc_stage_scope s(ctx, "short_combine");
        rc = msm_reduce_combine_launch(ctx, triples, K, SH_BINS, spl, scratch, out_ext, ctx->short_cursors, SH_BINS,
                                       poison);
    }
    if (rc != VMPC_OK) {
        VMPC_IGNORE(hipMemsetAsync(ctx->short_cursors, 0, 2 * SH_BINS * sizeof(uint32_t), st));
        ctx->short_path = 0;       // this context's device cannot run the path: general path from now on
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "short-commitment path: launch failed; path switched off for this context");
        return rc;
    }
    ctx->short_ready = true;      // (the combine launcher sets its kernel's LDS limit while this is still false)
    return VMPC_OK;
}
