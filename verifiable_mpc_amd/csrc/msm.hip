// Pippenger multi-scalar multiplication over Ed25519 for gfx950.
//
// Replaces the hot loop of the reference's Pedersen vector commitment,
//     prod = list_mul([g[i] ** _int(x_i) for i, x_i in enumerate(x)])
//     verifiable_mpc/ac20/pivot.py:143-144
// (n independent 253-bit double-and-add ladders followed by a product tree) by a
// windowed bucket method.  The group element produced is the same; only its affine
// normal form is defined as output.
//
// Pipeline (all on one stream, no host round trip):
//   prep     affine (x,y) -> niels (y-x, y+x, 2dxy), one 128-B line per point, coalesced 16-B accesses
//   recode, hist1, scan, part1, fine, plan   (msm_sort.hip: signed digits, two-level bucket sort,
//            segment task table)
//   bucket   one lane per segment: mixed additions (7M); finish: LDS tree for split buckets
//   reduce   per window: chunked running sums + LDS tree -> sum_b b*B_b partials
//   final    window sums, Horner over windows (c doublings each), one inversion -> affine
//
// HBM traffic is dominated by the 96-B niels gathers (W per term) and the 4-B sorted
// indices; the algorithmic bytes of SURVEY.md 8d are 96 B per term (32 B scalar + 64 B
// point).  Arithmetic is 32x32->64 integer multiply-add; no MFMA.
#include <stdlib.h>

#include "common.h"
#include "fe25519.h"
#include "fr.h"
#include "ge25519.h"
#include "scan.h"
#include "quad.h"
#include "msm_sort.h"
#include "ptio.h"

// niels record stride in 32-bit words: 24 = packed 96 B, 32 = one 128-B line per gather
#ifndef MSM_NIELS_STRIDE
#define MSM_NIELS_STRIDE 24
#endif
#ifndef MSM_REDUCE_WAVES
#define MSM_REDUCE_WAVES 2
#endif

// ---- prep: affine -> niels ------------------------------------------------------------
__global__ void __launch_bounds__(MSM_BLOCK)
k_msm_prep(const uint32_t *__restrict__ aff, size_t n_main, const uint32_t *__restrict__ aff_extra,
           size_t n_total, uint32_t *__restrict__ niels) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const uint32_t *src = (i < n_main) ? aff + 16 * i : aff_extra + 16 * (i - n_main);
    ge_aff a;
    a.x = fe_ld8(src);
    a.y = fe_ld8(src + 8);
    ge_niels q = ge_niels_from_affine(a);
    niels_st_line(niels + NIELS_WORDS * i, q);
}

// ---- bucket accumulation, segment-balanced (task table: msm_sort.hip) -------------------------------
__device__ __forceinline__ ge_niels niels_ld(const uint32_t *niels, uint32_t e) {
    return niels_ld_line(niels + NIELS_WORDS * (size_t)(e & 0x7fffffffu));
}

// one lane = one segment of <= MSM_SEG sorted entries
#ifndef MSM_IDX_BATCH
#define MSM_IDX_BATCH 4
#endif
#ifndef MSM_BUCKET_WAVES
#define MSM_BUCKET_WAVES 4
#endif
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, BLOCK == MSM_BLOCK ? MSM_BUCKET_WAVES : 1)
k_msm_bucket(const uint32_t *__restrict__ niels, const uint32_t *__restrict__ sorted,
             const uint32_t *__restrict__ starts, const uint32_t *__restrict__ counts,
             const uint32_t *__restrict__ nseg, const uint32_t *__restrict__ seg_starts,
             const uint2 *__restrict__ tasks, const uint32_t *__restrict__ n_tasks, int nb1, int seg, int balanced,
             uint32_t *__restrict__ buckets, uint32_t *__restrict__ partial) {
    // grid-stride over the task table: a launch with fewer workgroups than tasks / 256 (persistent form,
    // msm_accumulate) leaves register-file room on every SIMD for other streams' kernels
    // (measured and dropped, round 4: the stage as 2 / 4 / 8 launches over slices of the task table, to give other
    // streams' kernels a way in at every slice boundary - 0.87 / 0.87 / 0.90 ms per commitment against 0.875)
    const uint32_t n_live = *n_tasks;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_live; t += gridDim.x * blockDim.x) {
    uint2 tk = tasks[t];
    uint32_t ci = tk.x, sidx = tk.y;
    const uint32_t ns = nseg[ci];
    uint32_t lo, len;
    msm_seg_range(counts[ci], ns, sidx, (uint32_t)seg, balanced, lo, len);
    lo += starts[ci];
    ge_ext acc = ge_ext_identity();
    // Sorted indices in groups of MSM_IDX_BATCH.  A lane walks its own segment, so the lanes' 4-byte index loads
    // are >= 256 bytes apart - one cache line per lane - and with 2^18 lanes in flight (64 MB of lines in use
    // against 32 MB of L2) that line has usually left the L2 by the time the lane comes back for its next
    // index: one index at a time, the kernel moved 3.7 GB per launch over the fabric for 2.2 GB of table lines
    // and 0.07 GB of indices (profiles/r03a_pmc_summary.json).  Asking for a group's indices back to back lets
    // them share the line the first request brings in: 3.0 GB (scripts/pmc_ab.sh).  The kernel's TIME does not
    // change (integer-ALU bound; 8 at a time is slower: registers) - this only stops wasting fabric bandwidth
    // that the sort kernels of the next commitment in flight can use.
    // The next table entry is requested one addition ahead, but the compiler hoists its sign selection to right
    // behind the load, so in effect the gather latency is hidden by the 4 waves per SIMD, not by this
    // "prefetch".  A hand-scheduled version (inline-asm loads + manual s_waitcnt after the 7 multiplications,
    // 140 VGPRs, 3 waves) measured the same 0.61 ms for this kernel and 0.76 instead of 0.85 ms on the 2 GiB
    // fixed-base table: not worth carrying loads the compiler cannot see.  Staging the gather through LDS with
    // LDS-DMA (global_load_lds_dwordx4 into a [piece][lane] stage, no VGPR cost) measured 1.07 ms.
    // Initialising the accumulator from the first term (1M instead of 7M) lost to register pressure.
    // Round 4 settled what the kernel is bound by (profiles/r04_probes/bucket_prefetch_and_occupancy.txt): a variant with
    // the next entry REALLY prefetched (the eight loads as inline assembly between sched_barriers right after the third
    // multiplication, awaited after the seventh: 157 VGPRs, three waves) and a variant that touches the next line
    // early both take the same 660 us as this kernel; confined to 3 / 2 / 1 workgroups per CU all of them take 735 /
    // 760-770 / 880 us, with or without prefetch.  It is bound by vector-ALU issue, not by the gather's latency:
    // one wave per SIMD already issues 75 % of what four do.
    uint32_t ev[MSM_IDX_BATCH];
#pragma unroll
    for (int k = 0; k < MSM_IDX_BATCH; k++) ev[k] = sorted[lo + ((uint32_t)k < len ? (uint32_t)k : len - 1)];
    uint32_t e = ev[0];
    ge_niels q = niels_ld(niels, e);
    for (uint32_t j0 = 0; j0 < len; j0 += MSM_IDX_BATCH) {
        uint32_t en_v[MSM_IDX_BATCH];
#pragma unroll
        for (int k = 0; k < MSM_IDX_BATCH; k++) {
            const uint32_t jj = j0 + MSM_IDX_BATCH + (uint32_t)k;
            en_v[k] = sorted[lo + (jj < len ? jj : len - 1)];
        }
#pragma unroll
        for (int k = 0; k < MSM_IDX_BATCH; k++) {
            if (j0 + (uint32_t)k < len) {
                const uint32_t en = k + 1 < MSM_IDX_BATCH ? ev[k + 1 < MSM_IDX_BATCH ? k + 1 : 0] : en_v[0];
                ge_niels qn = niels_ld(niels, en);
                acc = ge_madd(acc, ge_niels_select_neg(q, (e >> 31) != 0));
                e = en;
                q = qn;
            }
        }
#pragma unroll
        for (int k = 0; k < MSM_IDX_BATCH; k++) ev[k] = en_v[k];
    }
    if (ns == 1)
        ext_st(buckets + EXT_WORDS * msm_bucket_slot(ci, nb1), acc);
    else
        ext_st(partial + EXT_WORDS * (size_t)(seg_starts[ci] + sidx), acc);
    }
}

// buckets that took several segments.  Lightly split ones (<= MSM_FINISH_SERIAL partial sums,
// e.g. the under-full top window) are summed by one lane each; heavily split ones (skewed
// scalars) by a workgroup-wide LDS tree.

// One launch for both: workgroups [0, gridDim.x / 2) sum the lightly split buckets, the other half the heavily split
// ones (two launches cost a dispatch each, 5 us of the ~250 us of a late prover round).
__global__ void __launch_bounds__(MSM_BLOCK)
k_msm_bucket_finish(const uint32_t *__restrict__ heavy_list, const uint32_t *__restrict__ ctrl,
                    const uint32_t *__restrict__ nseg, const uint32_t *__restrict__ seg_starts,
                    const uint32_t *__restrict__ partial, int nb1, uint32_t *__restrict__ buckets) {
    __shared__ uint32_t lds[MSM_BLOCK * EXT_WORDS];
    const uint32_t half = gridDim.x / 2;
    const uint32_t n_heavy = ctrl[0];
    if (blockIdx.x < half) {
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n_heavy; h += half * blockDim.x) {
        uint32_t ci = heavy_list[h];
        uint32_t ns = nseg[ci];
        if (ns > MSM_FINISH_SERIAL) continue;
        const uint32_t *src = partial + EXT_WORDS * (size_t)seg_starts[ci];
        ge_ext acc = ext_ld(src);
        for (uint32_t j = 1; j < ns; j++) acc = ge_add(acc, ext_ld(src + EXT_WORDS * (size_t)j));
        ext_st(buckets + EXT_WORDS * msm_bucket_slot(ci, nb1), acc);
    }
    return;
    }
    if (ctrl[4] == 0) return;                      // no heavily split bucket: nothing for the workgroup trees
    for (uint32_t h = blockIdx.x - half; h < n_heavy; h += half) {
        uint32_t ci = heavy_list[h];
        uint32_t ns = nseg[ci];
        if (ns <= MSM_FINISH_SERIAL) continue;          // wave-uniform: whole workgroup skips
        const uint32_t *src = partial + EXT_WORDS * (size_t)seg_starts[ci];
        ge_ext acc = ge_ext_identity();
        for (uint32_t j = threadIdx.x; j < ns; j += blockDim.x) acc = ge_add(acc, ext_ld(src + EXT_WORDS * (size_t)j));
        ext_st(lds + EXT_WORDS * threadIdx.x, acc);
        __syncthreads();
        uint32_t width = ns < MSM_BLOCK ? ns : MSM_BLOCK;   // lanes >= width hold the identity
        uint32_t stride = 1;
        while (stride < width) stride <<= 1;
        for (stride >>= 1; stride >= 1; stride >>= 1) {
            if (threadIdx.x < stride && threadIdx.x + stride < width)
                ext_st(lds + EXT_WORDS * threadIdx.x,
                       ge_add(ext_ld(lds + EXT_WORDS * threadIdx.x), ext_ld(lds + EXT_WORDS * (threadIdx.x + stride))));
            __syncthreads();
        }
        if (threadIdx.x == 0) ext_st(buckets + EXT_WORDS * msm_bucket_slot(ci, nb1), ext_ld(lds));
        __syncthreads();
    }
}

// ---- reduce: sum_b b * B_b per window ---------------------------------------------------
// thread = one chunk of `chunk_len` consecutive buckets; running sums inside the chunk,
// chunk offset by a short double-and-add, then an LDS tree over the workgroup.
// Tried in round 2 and dropped: replacing the per-lane ladder by a suffix scan of the chunk sums through LDS
// (8 Hillis-Steele steps) with the workgroup offsets 256 L g A_g left to the recombination kernel.  This kernel
// went from 160 to 130 us, the single-wave recombination from 240 to 340 us, and the step time with three
// commitments in flight did not move (1.12 ms): the VALU work saved here is not what bounds the pipeline.
// Also tried for the prover's latency (one or two bucket sets, every bucket its own lane): a QUAD of lanes per
// bucket (quad.h: 2 dependent multiplications per point operation instead of 8-9).  A/B on one box, compact
// prover: 23.6-24.5 ms with it, 23.3-24.4 ms without at N = 2^20; 6.6 / 6.6 ms at 2^15.  2^16 weights b * B_b of
// 15 bits are 1.5 M point operations - the stage is bound by that work, not by the length of one lane's chain.
__global__ void __launch_bounds__(MSM_BLOCK, MSM_REDUCE_WAVES)
k_msm_reduce(const uint32_t *__restrict__ buckets, const uint32_t *__restrict__ counts, int nb,
             int chunks, int chunk_len, int log2_chunk_len, int red_blocks,
             uint32_t *__restrict__ partials) {
    __builtin_amdgcn_s_setprio(3);   // latency chain: win issue arbitration against co-resident bucket waves
    __shared__ uint32_t lds[MSM_BLOCK * EXT_WORDS];
    const int w = blockIdx.y;
    const int chunk = blockIdx.x * blockDim.x + threadIdx.x;
    ge_ext contrib = ge_ext_identity();
    if (chunk < chunks) {
        const int lo = chunk * chunk_len;  // 0-based bucket index; bucket value = index + 1
        const uint32_t *bw = buckets + EXT_WORDS * ((size_t)w * nb + lo);
        const uint32_t *cw = counts + (size_t)w * (nb + 1) + lo + 1;   // empty buckets are never written
        ge_ext acc = ge_ext_identity(), sum = ge_ext_identity();
        for (int j = chunk_len - 1; j >= 0; j--) {
            if (cw[j]) acc = ge_add(acc, ext_ld(bw + EXT_WORDS * j));
            sum = ge_add(sum, acc);
        }
        // sum = sum_j (j+1) B_{lo+j}; add lo * acc where lo = chunk * 2^log2_chunk_len
        if (chunk != 0) {
            ge_ext base = acc;
            for (int k = 0; k < log2_chunk_len; k++) base = ge_dbl(base);
            ge_ext r = ge_ext_identity();
            int top = 31 - __clz(chunk);
            for (int k = top; k >= 0; k--) {
                r = ge_dbl(r);
                if ((chunk >> k) & 1) r = ge_add(r, base);
            }
            sum = ge_add(sum, r);
        }
        contrib = sum;
    }
    ext_st(lds + EXT_WORDS * threadIdx.x, contrib);
    __syncthreads();
    for (int stride = MSM_BLOCK / 2; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride) {
            ge_ext a = ext_ld(lds + EXT_WORDS * threadIdx.x);
            ge_ext b = ext_ld(lds + EXT_WORDS * (threadIdx.x + stride));
            ext_st(lds + EXT_WORDS * threadIdx.x, ge_add(a, b));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        ext_st(partials + EXT_WORDS * ((size_t)w * red_blocks + blockIdx.x), ext_ld(lds));
}

// ---- final: window sums, Horner, normalise -----------------------------------------------
// The Horner recombination  acc = sum_w 2^(c*w) R_w  is a chain of ~253 dependent point
// doublings: pure latency.  Each doubling / addition has two levels of four independent
// field multiplications, so a QUAD of lanes computes one point operation: lane q of the quad
// does the q-th product of each level and the four results are exchanged with DPP quad_perm
// broadcasts (VALU, no LDS).  Per point operation the critical path is 2 field
// multiplications instead of 8-9.
// The accumulator stays LANE-DISTRIBUTED (quad.h): lane q of every quad holds coordinate q of
// (X, Y, Z, T).  Second level of both operations: lane 0: E*F = X3, lane 1: G*H = Y3, lane 2: G*F = Z3,
// lane 3: E*H = T3 - two-way operand selections, and the products land where the next operation reads
// them.  Sums stay lazy as in ge25519.h (only F is carried).  448 instructions per doubling instead
// of 733 for the replicated form with four-way picks and carried sums.
__global__ void __launch_bounds__(64)
k_msm_final(const uint32_t *__restrict__ partials, int W, int red_blocks, int c,
            uint32_t *__restrict__ out_ext, uint32_t *__restrict__ out_aff,
            uint32_t *done_counter, uint32_t *done_flag, uint32_t done_seq) {
    __builtin_amdgcn_s_setprio(3);   // latency chain: win issue arbitration against co-resident bucket waves
    __shared__ uint32_t lds[64 * EXT_WORDS];
    // block k = commitment k of a batch: its W windows, its own 128-byte / 64-byte output slot
    partials += (size_t)EXT_WORDS * blockIdx.x * W * red_blocks;
    if (out_ext) out_ext += 32 * blockIdx.x;
    if (out_aff) out_aff += 16 * blockIdx.x;
    // phase 1: window sums.  lpw lanes share a window (strided partials), then a short tree.
    int lpw = 1;
    while (lpw * 2 * W <= 64 && lpw * 2 <= red_blocks) lpw *= 2;
    {
        const int w = threadIdx.x / lpw, sub = threadIdx.x % lpw;
        ge_ext r = ge_ext_identity();
        if (w < W)
            for (int j = sub; j < red_blocks; j += lpw)
                r = ge_add(r, ext_ld(partials + EXT_WORDS * ((size_t)w * red_blocks + j)));
        ext_st(lds + EXT_WORDS * threadIdx.x, r);
        __syncthreads();
        for (int stride = lpw / 2; stride >= 1; stride >>= 1) {
            if (w < W && sub < stride)
                ext_st(lds + EXT_WORDS * threadIdx.x,
                       ge_add(ext_ld(lds + EXT_WORDS * threadIdx.x), ext_ld(lds + EXT_WORDS * (threadIdx.x + stride))));
            __syncthreads();
        }
        ge_ext tot = ext_ld(lds + EXT_WORDS * (w < W ? w * lpw : 0));
        __syncthreads();
        // window w's sum, in cached form for the cooperative additions, at slot w
        if (w < W && sub == 0) {
            fe_st(lds + EXT_WORDS * w, fe_sub(tot.Y, tot.X));                       // lane 0's partner
            fe_st(lds + EXT_WORDS * w + FE_LIMBS, fe_add(tot.Y, tot.X));            // lane 1's
            fe_st(lds + EXT_WORDS * w + 2 * FE_LIMBS, fe_mul(tot.T, fe_const_d2()));  // lane 2's
            fe_st(lds + EXT_WORDS * w + 3 * FE_LIMBS, fe_dbl(tot.Z));               // lane 3's
        }
    }
    __syncthreads();
    // every quad of the wave runs the same chain redundantly (keeps EXEC full for DPP)
    const int q = threadIdx.x & 3;
    fe P = (q == 1 || q == 2) ? fe_one() : fe_zero();          // identity (0 : 1 : 1 : 0), distributed
    for (int k = W - 1; k >= 0; k--) {
        if (k != W - 1)
            for (int j = 0; j < c; j++) P = quadD_dbl(P, q);
        P = quadD_add_cached(P, fe_ld(lds + EXT_WORDS * k + FE_LIMBS * q), q);
    }
    ge_ext acc;
    acc.X = quad_perm<0x00>(P);
    acc.Y = quad_perm<0x55>(P);
    acc.Z = quad_perm<0xaa>(P);
    acc.T = quad_perm<0xff>(P);
    if (threadIdx.x == 0) {
        if (out_ext) ext_st8(out_ext, acc);        // public: packed 128-byte X||Y||Z||T
        if (out_aff) {
            ge_aff a = ge_ext_to_affine(acc);
            fe_st8(out_aff, a.x);
            fe_st8(out_aff + 8, a.y);
        }
        if (done_flag) vmpc_publish_done(done_counter, done_flag, done_seq);   // a prover round queued ahead (prover.hip)
    }
}

// ---- sum of m extended points in order (multi-GPU combine) --------------------------------
// k independent sums in one launch (block j = sum j): point i of sum j sits at pts + 32 * (i * k + j),
// i.e. the layout an all-gather of k points per rank produces
__global__ void __launch_bounds__(64)
k_points_sum(const uint32_t *__restrict__ pts, size_t m, size_t k, uint32_t *__restrict__ out_ext,
             uint32_t *__restrict__ out_aff) {
    if (threadIdx.x != 0) return;
    const size_t j = blockIdx.x;
    ge_ext acc = ge_ext_identity();
    for (size_t i = 0; i < m; i++) acc = ge_add(acc, ext_ld8(pts + 32 * (i * k + j)));   // packed inputs
    if (out_ext) ext_st8(out_ext + 32 * j, acc);
    if (out_aff) {
        ge_aff a = ge_ext_to_affine(acc);
        fe_st8(out_aff + 16 * j, a.x);
        fe_st8(out_aff + 16 * j + 8, a.y);
    }
}

// ---- ALU ceiling probe (bench.py): the bucket stage's inner operation with no memory traffic ---
__global__ void __launch_bounds__(MSM_BLOCK)
k_madd_rate(const uint32_t *__restrict__ seed, int iters, uint32_t *__restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    ge_aff a;
    a.x = fe_ld8(seed);
    a.y = fe_ld8(seed + 8);
    a.x.v[0] ^= (uint32_t)i & 0xffu;          // lanes differ; the values need not be on the curve
    ge_niels q = ge_niels_from_affine(a);
    ge_ext p = ge_ext_identity();
    for (int k = 0; k < iters; k++) {
        p = ge_madd(p, q);
        q.t2d.v[0] ^= (uint32_t)k & 1u;
    }
    fe s = fe_add(fe_add(p.X, p.Y), fe_add(p.Z, p.T));
    if (s.v[0] == seed[31]) fe_st8(sink, s);   // seed[31] = 2^32-1: never a reduced limb; keeps the chain live
}

extern "C" int vmpc_ed25519_madd_rate(vmpc_ctx *ctx, int iters, double *madds_per_second) {
    if (!ctx || !madds_per_second || iters < 1) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_CHECK(vmpc_ws_reserve(ctx, 512));
    uint32_t *buf = (uint32_t *)vmpc_ws_take(ctx, 256);
    uint32_t host[32] = {0};
    host[0] = 9;
    host[8] = 5;
    host[31] = 0xffffffffu;
    VMPC_HIP_CHECK(hipMemcpyAsync(buf, host, sizeof host, hipMemcpyHostToDevice, ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    hipEvent_t e0, e1;
    VMPC_HIP_CHECK(hipEventCreate(&e0));
    VMPC_HIP_CHECK(hipEventCreate(&e1));
    const unsigned blocks = 8u * (unsigned)ctx->cu_count;
    k_madd_rate<<<blocks, MSM_BLOCK, 0, ctx->stream>>>(buf, 8, buf + 32);     // warm-up
    VMPC_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    k_madd_rate<<<blocks, MSM_BLOCK, 0, ctx->stream>>>(buf, iters, buf + 32);
    VMPC_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    VMPC_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    VMPC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms <= 0.f) return VMPC_E_HIP;
    *madds_per_second = (double)blocks * MSM_BLOCK * (double)iters / (ms * 1e-3);
    return VMPC_OK;
}

static const msm_modulus ED25519_L = {VMPC_FR_L};

extern "C" int vmpc_ed25519_msm_plan(vmpc_ctx *ctx, size_t n, int *c_bits, int *windows) {
    if (!ctx || !c_bits || !windows || n == 0) return VMPC_E_INVAL;
    msm_plan p;
    msm_make_plan(ctx, n, 0, 253, p, &ED25519_L);
    *c_bits = p.c;
    *windows = p.W;
    return VMPC_OK;
}

// ---- validation ---------------------------------------------------------------------------
__global__ void __launch_bounds__(MSM_BLOCK)
k_points_validate(const uint32_t *__restrict__ aff, size_t n, unsigned long long *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t wx[8], wy[8];
    load_u32x8(wx, aff + 16 * i);
    load_u32x8(wy, aff + 16 * i + 8);
    ge_aff a;
    a.x = fe_unpack(wx);
    a.y = fe_unpack(wy);
    bool ok = fe8_is_canonical(wx) && fe8_is_canonical(wy) && ge_aff_on_curve(a);
    if (!ok) atomicAdd(bad, 1ull);
}


// bucket accumulation -> finish -> reduce -> recombination over a sorted task table; `entries` is
// the niels array the sorted indices refer to (the call's own prepared points, or a fixed-base table)
static int msm_accumulate(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, const uint32_t *entries, void *out_ext,
                          void *out_affine) {
    const int batch = p.W / p.period;        // commitments sharing this pass (1 unless vmpc_msm_table_batch_dev)
    hipStream_t st = ctx->stream;
    // Phase pipelining: with a shared bucket stream the bucket kernels of a pipeline's contexts run one at a time,
    // back to back, and leave room on every SIMD (persistent launch of bucket_wgs_per_cu workgroups per CU) for the
    // sort of the next pass and the reduction / recombination of the previous one on the contexts' own streams.
    hipStream_t bst = ctx->bucket_stream ? ctx->bucket_stream : st;
    if (ctx->bucket_stream) {
        VMPC_HIP_CHECK(hipEventRecord(ctx->ev_sorted, st));
        VMPC_HIP_CHECK(hipStreamWaitEvent(bst, ctx->ev_sorted, 0));
    }
    {
        ctx->stage_stream = ctx->bucket_stream;
        vmpc_stage_scope s(ctx, "msm_bucket");
        unsigned grid = (unsigned)((w.t_max + MSM_BLOCK - 1) / MSM_BLOCK);
        if (ctx->bucket_wgs_per_cu > 0 && (unsigned)(ctx->bucket_wgs_per_cu * ctx->cu_count) < grid)
            grid = (unsigned)(ctx->bucket_wgs_per_cu * ctx->cu_count);
        if (ctx->bucket_block == 1024) {
            // experiment (VMPC_EXPERIMENTAL=1 VMPC_BUCKET_BLOCK=1024): one workgroup fills a CU, so a retiring one frees
            // a whole CU at once - room for the 1024-thread sort workgroups of the other streams
            k_msm_bucket<1024><<<(grid + 3) / 4, 1024, 0, bst>>>(
                entries, w.sorted, w.starts, w.counts, w.nseg, w.seg_starts, w.tasks, w.ctrl + 1, p.nb1,
                (int)msm_seg_len(p), p.balanced, w.buckets, w.seg_partial);
        } else {
            k_msm_bucket<MSM_BLOCK><<<grid, MSM_BLOCK, 0, bst>>>(
                entries, w.sorted, w.starts, w.counts, w.nseg, w.seg_starts, w.tasks, w.ctrl + 1, p.nb1,
                (int)msm_seg_len(p), p.balanced, w.buckets, w.seg_partial);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            ctx->stage_stream = nullptr;
            VMPC_HIP_CHECK(e);
        }
    }
    ctx->stage_stream = nullptr;
    if (ctx->bucket_stream) {
        VMPC_HIP_CHECK(hipEventRecord(ctx->ev_bucketed, bst));
        VMPC_HIP_CHECK(hipStreamWaitEvent(st, ctx->ev_bucketed, 0));
    }
    {
        vmpc_stage_scope s(ctx, "msm_bucket_finish");
        k_msm_bucket_finish<<<4 * ctx->cu_count, MSM_BLOCK, 0, st>>>(w.heavy_list, w.ctrl, w.nseg,
                                                                    w.seg_starts, w.seg_partial, p.nb1,
                                                                    w.buckets);
        VMPC_KERNEL_CHECK();
    }
    // short chunks (latency path: one commitment alone, the prover's rounds): weights from the quad tree
    const bool tree = ctx->reduce_tree && msm_reduce_tree_fits(p);
    // ... and with one bucket set per commitment (a 16-row table) its second kernel writes the results itself
    const int tree_parts = tree ? msm_reduce_tree_split(p) : 1;
    const bool tree_is_final = tree && p.period == 1 && out_ext && !out_affine && tree_parts == 1;
    {
        vmpc_stage_scope s(ctx, "msm_reduce");
        if (tree) {
            VMPC_CHECK(msm_reduce_tree(ctx, p, w, st, tree_is_final ? out_ext : nullptr));
        } else {
            k_msm_reduce<<<dim3(p.red_blocks, p.W), MSM_BLOCK, 0, st>>>(
                w.buckets, w.counts, p.nb, p.chunks, p.chunk_len, msm_ilog2(p.chunk_len), p.red_blocks, w.partials);
            VMPC_KERNEL_CHECK();
        }
    }
    if (!tree_is_final) {
        vmpc_stage_scope s(ctx, "msm_final");
        k_msm_final<<<batch, 64, 0, st>>>(w.partials, p.period, tree ? tree_parts : p.red_blocks, p.c, (uint32_t *)out_ext,
                                          (uint32_t *)out_affine, ctx->d_status + VMPC_ST_WORDS, ctx->done_flag_dev,
                                          ctx->done_seq);
        ctx->done_flag_dev = nullptr;
        VMPC_KERNEL_CHECK();
    }
    return VMPC_OK;
}


extern "C" int vmpc_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *affine_points, size_t n,
                            const void *extra_scalars, const void *extra_affine_points,
                            size_t n_extra, void *out_ext, void *out_affine) {
    if (!ctx || (n && (!scalars || !affine_points)) || (n_extra && (!extra_scalars || !extra_affine_points)))
        return VMPC_E_INVAL;
    if (!out_ext && !out_affine) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    size_t n_total = n + n_extra;
    if (n_total == 0) {
        // empty product = identity (pivot.list_mul's `initial`, pivot.py:28)
        uint32_t id_ext[32] = {0}, id_aff[16] = {0};
        id_ext[8] = 1; id_ext[16] = 1; id_aff[8] = 1;
        if (out_ext) VMPC_HIP_CHECK(hipMemcpyAsync(out_ext, id_ext, 128, hipMemcpyHostToDevice, st));
        if (out_affine) VMPC_HIP_CHECK(hipMemcpyAsync(out_affine, id_aff, 64, hipMemcpyHostToDevice, st));
        VMPC_HIP_CHECK(hipStreamSynchronize(st));
        return VMPC_OK;
    }
    if (n_total >= (1ull << 31) / 16) return VMPC_E_INVAL;  // index / offset width
    msm_plan p;
    msm_make_plan(ctx, n, n_extra, 253, p, &ED25519_L);
    msm_ws w;
    msm_layout(p, w, nullptr, NIELS_WORDS * 4, EXT_WORDS * 4);
    VMPC_CHECK(vmpc_ws_reserve(ctx, w.total));
    msm_layout(p, w, (char *)ctx->ws, NIELS_WORDS * 4, EXT_WORDS * 4);

    {
        vmpc_stage_scope s(ctx, "msm_prep");
        k_msm_prep<<<(unsigned)((n_total + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, st>>>(
            (const uint32_t *)affine_points, n, (const uint32_t *)extra_affine_points, n_total, w.entries);
        VMPC_KERNEL_CHECK();
    }
    VMPC_CHECK(msm_sort_stage(ctx, p, w, scalars, n, extra_scalars, ED25519_L));
    return msm_accumulate(ctx, p, w, w.entries, out_ext, out_affine);
}

// ---- fixed-base tables -----------------------------------------------------------------------
// A generator vector that serves many commitments (the CRS of pivot.py:139-145: g, h, k never
// change between proofs) is expanded once into r rows  T[rho][i] = 2^(256 rho / r) * P_i  in niels
// form, r in {1, 2, 4, 8, 16}.  The 16 digits of a scalar then fall into 16 / r bucket SETS: digit w
// adds T[w / (16/r)][i] into bucket |digit| of set w % (16/r), and each set's flattened digit row
// [rho][i] is sorted as one window whose indices ARE table positions.  What remains of the window
// recombination is a Horner chain over the 16 / r sets: (16/r - 1) * 16 doublings instead of 240,
// none at all for r = 16; the bucket reduction shrinks by the same factor and no per-call point
// preparation is left.  r trades table size (r * 128 bytes per generator) against that chain: the
// bucket stage gathers table entries at random, so the table should stay Infinity-Cache resident
// (256 MiB) - PointVector.precompute picks r accordingly.
// The digit width c need not be the 16 bits the rows are spaced for: any c dividing 256 / r works, window w
// (of 256 / c) then uses row w / (256 / (c r)) and set w % (256 / (c r)).  c = 16 has the fewest entries and is
// what every size uses; narrower digits (vmpc_ctx_set_window 4 or 8) were tried for SHORT commitments, whose
// time is the latency of reducing 2^15 buckets per set - they lose: 2^7 buckets with 2^8+ entries each turn the
// bucket stage into long serial runs (N = 2^12 prover: 8.0 ms at c = 8 against 6.1 ms at c = 16).
bool msm_short_fits(const vmpc_ctx *ctx, size_t table_n, size_t table_extra, size_t m, int rows, int c, int K,
                    const void *out_ext, const void *out_affine);                             // msm_short.hip
int msm_short_batch(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, const void *const *scalars,
                    size_t m, const void *const *extra_scalars, int K, void *out_ext, const msm_modulus &modulus);

static int msm_table_window(const vmpc_ctx *ctx) {
    const int o = ctx->window_override;      // vmpc_ctx_set_window / VMPC_MSM_WINDOW: honoured when it divides 16
    return (o == 4 || o == 8 || o == 16) ? o : 16;
}
// rows = MSM_WIDE_ROWS (13) is the WIDE-WINDOW table (round 6): rows spaced 20 bits, T[r][i] = 2^(20 r) * P_i, one set
// of 2^19 buckets per commitment, 13 mixed additions per term instead of 16 and no recombination; its row stride is a
// whole number of 8192-position sort chunks (msm_sort.h).  The other row counts are spaced 256 / rows bits.
static size_t msm_table_stride(size_t n_points, int rows = 16) {
    return rows == MSM_WIDE_ROWS ? (n_points + 8191) & ~(size_t)8191 : (n_points + 7) & ~(size_t)7;
}
static bool msm_table_rows_ok(int rows) {
    return rows == 1 || rows == 2 || rows == 4 || rows == 8 || rows == 16 || rows == MSM_WIDE_ROWS;
}
static int msm_table_row_bits(int rows) { return rows == MSM_WIDE_ROWS ? MSM_WIDE_C : 256 / rows; }

__global__ void __launch_bounds__(MSM_BLOCK, 2)
k_msm_table_build(const uint32_t *__restrict__ aff, size_t n_main, const uint32_t *__restrict__ aff_extra,
                  size_t n_total, size_t stride, int rows, size_t col_begin, uint32_t *__restrict__ table) {
    size_t i = col_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // columns [col_begin, stride)
    if (i >= stride) return;
    if (i >= n_total) {     // padding columns are never referenced (their digits are zero); keep them defined
        ge_niels z;
        z.ymx = fe_one();
        z.ypx = fe_one();
        z.t2d = fe_zero();
        for (int r = 0; r < rows; r++) niels_st_line(table + NIELS_WORDS * ((size_t)r * stride + i), z);
        return;
    }
    const uint32_t *src = (i < n_main) ? aff + 16 * i : aff_extra + 16 * (i - n_main);
    ge_aff a;
    a.x = fe_ld8(src);
    a.y = fe_ld8(src + 8);
    niels_st_line(table + NIELS_WORDS * i, ge_niels_from_affine(a));
    ge_ext q = ge_ext_from_affine(a);
    const int dbl_per_row = rows == MSM_WIDE_ROWS ? MSM_WIDE_C : 256 / rows;
    // Rows 1 .. rows-1 need the AFFINE form of 2^(k rho) P: one inversion for all of them (Montgomery's
    // trick).  Pass 1 parks (X, Y, Z, Z_1 ... Z_rho) of row rho in the row's own 128-byte slot; pass 2 walks
    // back with the running inverse and overwrites the slot with the niels entry.
    fe run = fe_one();
    for (int r = 1; r < rows; r++) {
        for (int k = 0; k < dbl_per_row; k++) q = ge_dbl(q);
        run = fe_mul(run, q.Z);
        uint32_t *slot = table + NIELS_WORDS * ((size_t)r * stride + i);
        fe_st8(slot, q.X);
        fe_st8(slot + 8, q.Y);
        fe_st8(slot + 16, q.Z);
        fe_st8(slot + 24, run);
    }
    if (rows > 1) {
        fe inv = fe_inv(run);
        for (int r = rows - 1; r >= 1; r--) {
            uint32_t *slot = table + NIELS_WORDS * ((size_t)r * stride + i);
            const fe Z = fe_ld8(slot + 16);
            const fe prev = r > 1 ? fe_ld8(table + NIELS_WORDS * ((size_t)(r - 1) * stride + i) + 24) : fe_one();
            const fe zi = fe_mul(inv, prev);
            inv = fe_mul(inv, Z);
            ge_aff b;
            b.x = fe_canon(fe_mul(fe_ld8(slot), zi));
            b.y = fe_canon(fe_mul(fe_ld8(slot + 8), zi));
            niels_st_line(slot, ge_niels_from_affine(b));
        }
    }
}

extern "C" int vmpc_msm_table_bytes(size_t n, size_t n_extra, int rows, size_t *bytes) {
    if (!bytes || n + n_extra == 0 || n + n_extra > ((size_t)1 << 26) || !msm_table_rows_ok(rows)) return VMPC_E_INVAL;
    *bytes = (size_t)rows * msm_table_stride(n + n_extra, rows) * NIELS_WORDS * 4;
    return VMPC_OK;
}

extern "C" int vmpc_msm_table_build_dev(vmpc_ctx *ctx, const void *affine_points, size_t n,
                                        const void *extra_affine_points, size_t n_extra, int rows, void *table) {
    if (!ctx || !table || (n && !affine_points) || (n_extra && !extra_affine_points) || n + n_extra == 0 ||
        n + n_extra > ((size_t)1 << 26) || !msm_table_rows_ok(rows))
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t stride = msm_table_stride(n + n_extra, rows);
    vmpc_stage_scope s(ctx, "msm_table_build");
    k_msm_table_build<<<(unsigned)((stride + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
        (const uint32_t *)affine_points, n, (const uint32_t *)extra_affine_points, n + n_extra, stride, rows, 0,
        (uint32_t *)table);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

// the extras' columns (and the padding) of a table whose generator columns another kernel fills (fold_jump.hip)
int vmpc_msm_table_build_extras(vmpc_ctx *ctx, size_t n, const void *extra_affine_points, size_t n_extra, int rows,
                                void *table) {
    const size_t stride = msm_table_stride(n + n_extra, rows);
    k_msm_table_build<<<(unsigned)((stride - n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
        nullptr, n, (const uint32_t *)extra_affine_points, n + n_extra, stride, rows, n, (uint32_t *)table);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

// The wide-window table: K commitments = K digit rows of 13 * stride positions, each its own set of 2^19 buckets
// (plan: c = 20, period = 1).  Sorted by the same two-level LDS sort (1024 coarse bins x 512 buckets), accumulated,
// finished and reduced by the same kernels; no recombination (k_msm_final only adds the reduction's partial sums).
static int msm_table_batch_wide(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra,
                                const void *const *scalars, size_t m, const void *const *extra_scalars, int K,
                                void *out_ext, void *out_affine) {
    const size_t stride = msm_table_stride(table_n + table_extra, MSM_WIDE_ROWS);
    if (stride > ((size_t)1 << 22)) return VMPC_E_INVAL;          // column + 9 fine bits + sign in a 32-bit entry
    msm_plan p;
    p.n_main = p.n_total = (size_t)MSM_WIDE_ROWS * stride;
    p.n_extra = 0;
    p.scalar_bits = 253;
    p.c = MSM_WIDE_C;
    p.period = 1;
    p.W = K;
    p.top_row = -1;
    p.top_max_b = 0;
    p.wide = 1;
    p.row_stride = stride;
    p.chunks_per_row = (int)(stride / 8192);
    p.col_bits = 1;
    while (((size_t)1 << p.col_bits) < stride) p.col_bits++;
    msm_plan_geometry(ctx, p);
    msm_ws w;
    msm_layout(p, w, nullptr, 0, EXT_WORDS * 4);
    VMPC_CHECK(vmpc_ws_reserve(ctx, w.total));
    msm_layout(p, w, (char *)ctx->ws, 0, EXT_WORDS * 4);
    VMPC_CHECK(msm_recode_wide_batch(ctx, scalars, m, extra_scalars, K, table_n, table_extra, stride, (int32_t *)w.digits,
                                     p.n_pad, ED25519_L));
    VMPC_CHECK(msm_sort_digits(ctx, p, w));
    return msm_accumulate(ctx, p, w, (const uint32_t *)table, out_ext, out_affine);
}

// K commitments over the same tabulated generators in ONE pass: the K scalar vectors are recoded into K * (16 / rows)
// digit rows, sorted together, accumulated by one bucket launch, reduced by one launch (K x the lanes for the
// same chain length) and recombined by K workgroups side by side - so the latency chains of the reduction and
// the Horner recombination are paid once per batch, not once per commitment.  (A_i and B_i of a Protocol-4
// round, compressed_pivot.py:41-42, are such a pair; so are independent commitments queued by a prover.)
static int msm_table_batch(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                           const void *const *scalars, size_t m, const void *const *extra_scalars, int K,
                           void *out_ext, void *out_affine) {
    if (!ctx || !table || m > table_n || (m && !scalars) || table_n + table_extra == 0 || K < 1 || K > 16 ||
        table_n + table_extra > ((size_t)1 << 26) || !msm_table_rows_ok(rows) || (!out_ext && !out_affine))
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    for (int k = 0; k < K; k++)
        if (m && !scalars[k]) return VMPC_E_INVAL;
    // one bucket set per commitment over a short table: three launches instead of seventeen (msm_short.hip)
    if (msm_short_fits(ctx, table_n, table_extra, m, rows, msm_table_window(ctx), K, out_ext, out_affine)) {
        if (ctx->short_backoff > 0) ctx->short_backoff--;      // an overflow a few calls ago: general path for now
        else return msm_short_batch(ctx, table, table_n, table_extra, scalars, m, extra_scalars, K, out_ext, ED25519_L);
    }
    if (rows == MSM_WIDE_ROWS)
        return msm_table_batch_wide(ctx, table, table_n, table_extra, scalars, m, extra_scalars, K, out_ext, out_affine);
    const size_t stride = msm_table_stride(table_n + table_extra);
    // each of the 16 / rows bucket sets of a commitment is one row of rows * stride entries
    msm_plan p;
    p.n_main = p.n_total = (size_t)rows * stride;
    p.n_extra = 0;
    p.scalar_bits = 253;
    p.c = msm_table_window(ctx);
    const int windows = 256 / p.c;
    p.period = windows / rows;
    p.W = K * p.period;
    p.top_row = -1;
    p.top_max_b = 0;
    if (rows == 1) {          // prepared generators: every window is its own row, the top one included
        msm_plan q;
        msm_make_plan(ctx, p.n_total, 0, 253, q, &ED25519_L);
        if (q.c == p.c && q.W == p.period) {
            p.top_row = q.top_row;
            p.top_max_b = q.top_max_b;
        }
    }
    msm_plan_geometry(ctx, p);
    msm_ws w;
    msm_layout(p, w, nullptr, 0, EXT_WORDS * 4);
    VMPC_CHECK(vmpc_ws_reserve(ctx, w.total));
    msm_layout(p, w, (char *)ctx->ws, 0, EXT_WORDS * 4);
    for (int k = 0; k < K; k++)
        if (m && !scalars[k]) return VMPC_E_INVAL;
    VMPC_CHECK(msm_recode_rows_batch(ctx, scalars, m, extra_scalars, K, table_n, table_extra, stride, w.digits,
                                     (size_t)p.period * p.n_pad, p.c, windows, rows, ED25519_L));
    VMPC_CHECK(msm_sort_digits(ctx, p, w));
    return msm_accumulate(ctx, p, w, (const uint32_t *)table, out_ext, out_affine);
}

extern "C" int vmpc_msm_table_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                  const void *scalars, size_t m, const void *extra_scalars, void *out_ext,
                                  void *out_affine) {
    const void *sc[1] = {scalars}, *ex[1] = {extra_scalars};
    return msm_table_batch(ctx, table, table_n, table_extra, rows, sc, m, extra_scalars ? ex : nullptr, 1, out_ext,
                           out_affine);
}

extern "C" int vmpc_msm_table_batch_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                        const void *const *scalars, size_t m, const void *const *extra_scalars,
                                        int batch, void *out_ext, void *out_affine) {
    return msm_table_batch(ctx, table, table_n, table_extra, rows, scalars, m, extra_scalars, batch, out_ext, out_affine);
}

// ---- fixed-base batch: out_i = n_i * B for one base B (generator setup, circuit_sat_r1cs.py:64-70,81) --
// When only the group elements are wanted (affine, no reference representative), `h ** r_i` needs no
// 253-doubling ladder per element: a comb table  T[w][d-1] = d * 2^(8w) * B,  d = 1..128, w = 0..31
// (4096 niels entries, 512 KiB, L2 resident) turns every output into 32 mixed additions of signed
// 8-bit digits plus one inversion - 6x less work than the exact replay of vmpc_repeat_dev.
int vmpc_normalize_launch(vmpc_ctx *ctx, const void *proj, size_t n, void *out_affine);   // exact.hip

#define FB_C 8
#define FB_W 32
#define FB_D 128

__global__ void __launch_bounds__(64)
k_fb_bases(const uint32_t *__restrict__ base_aff, uint32_t *__restrict__ bases /*FB_W ext*/) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ge_aff a;
    a.x = fe_ld8(base_aff);
    a.y = fe_ld8(base_aff + 8);
    ge_ext q = ge_ext_from_affine(a);
    for (int w = 0; w < FB_W; w++) {
        ext_st(bases + EXT_WORDS * w, q);
        for (int k = 0; k < FB_C; k++) q = ge_dbl(q);
    }
}

__global__ void __launch_bounds__(MSM_BLOCK)
k_fb_table(const uint32_t *__restrict__ bases, uint32_t *__restrict__ table /*FB_W * FB_D niels*/) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_W * FB_D) return;
    const int w = t / FB_D, d = t % FB_D + 1;
    const ge_ext b = ext_ld(bases + EXT_WORDS * w);
    ge_ext r = ge_ext_identity();
    for (int k = 7; k >= 0; k--) {            // d <= 128: 8-bit left-to-right ladder
        r = ge_dbl(r);
        if ((d >> k) & 1) r = ge_add(r, b);
    }
    niels_st_line(table + NIELS_WORDS * t, ge_niels_from_affine(ge_ext_to_affine(r)));
}

__global__ void __launch_bounds__(MSM_BLOCK)
k_fb_apply(const uint32_t *__restrict__ table, const uint32_t *__restrict__ sc, size_t n, msm_modulus mod,
           uint32_t *__restrict__ status, uint32_t *__restrict__ out_proj) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8];
    load_u32x8(s, sc + 8 * i);
    {
        bool ge = true;
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            if (s[k] != mod.v[k]) {
                ge = s[k] > mod.v[k];
                break;
            }
        }
        if (ge) atomicAdd(&status[VMPC_ST_NONCANON], 1u);
    }
    ge_ext acc = ge_ext_identity();
    uint32_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < FB_W; w++) {
        uint32_t raw = ((s[w >> 2] >> (8 * (w & 3))) & 0xffu) + carry;
        int d;
        if (raw > FB_D) {
            d = (int)raw - 256;
            carry = 1;
        } else {
            d = (int)raw;
            carry = 0;
        }
        if (d != 0) {
            const uint32_t idx = (uint32_t)w * FB_D + (uint32_t)((d < 0 ? -d : d) - 1);
            acc = ge_madd(acc, ge_niels_select_neg(niels_ld_line(table + NIELS_WORDS * idx), d < 0));
        }
    }
    // scalars are < l < 2^253: the top digit is < 32, no carry leaves the last window
    // (X : Y : Z) goes out un-normalised: the inversion would be 265 of this lane's 489 multiplications;
    // vmpc_normalize_launch shares one among eight elements (Montgomery's trick, exact.hip)
    fe_st8(out_proj + 24 * i, acc.X);
    fe_st8(out_proj + 24 * i + 8, acc.Y);
    fe_st8(out_proj + 24 * i + 16, acc.Z);
}

extern "C" int vmpc_fixed_base_dev(vmpc_ctx *ctx, const void *base_affine, const void *scalars, size_t n,
                                   void *out_affine) {
    if (!ctx || !base_affine || (n && (!scalars || !out_affine))) return VMPC_E_INVAL;
    if (n == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t bases_bytes = vmpc_align((size_t)FB_W * EXT_WORDS * 4);
    const size_t table_bytes = vmpc_align((size_t)FB_W * FB_D * NIELS_WORDS * 4);
    const size_t proj_bytes = vmpc_align(n * 96);
    VMPC_CHECK(vmpc_ws_reserve(ctx, bases_bytes + table_bytes + proj_bytes + 1024));
    uint32_t *bases = (uint32_t *)vmpc_ws_take(ctx, bases_bytes);
    uint32_t *table = (uint32_t *)vmpc_ws_take(ctx, table_bytes);
    uint32_t *proj = (uint32_t *)vmpc_ws_take(ctx, proj_bytes);
    vmpc_stage_scope s(ctx, "fixed_base");
    k_fb_bases<<<1, 64, 0, st>>>((const uint32_t *)base_affine, bases);
    VMPC_KERNEL_CHECK();
    k_fb_table<<<(FB_W * FB_D + MSM_BLOCK - 1) / MSM_BLOCK, MSM_BLOCK, 0, st>>>(bases, table);
    VMPC_KERNEL_CHECK();
    k_fb_apply<<<(unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, st>>>(
        table, (const uint32_t *)scalars, n, ED25519_L, ctx->d_status, proj);
    VMPC_KERNEL_CHECK();
    return vmpc_normalize_launch(ctx, proj, n, out_affine);
}

extern "C" int vmpc_points_sum_dev(vmpc_ctx *ctx, const void *ext_points, size_t m, void *out_ext,
                                   void *out_affine) {
    if (!ctx || (m && !ext_points) || (!out_ext && !out_affine)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "points_sum");
    k_points_sum<<<1, 64, 0, ctx->stream>>>((const uint32_t *)ext_points, m, 1, (uint32_t *)out_ext,
                                           (uint32_t *)out_affine);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_points_sum_many_dev(vmpc_ctx *ctx, const void *ext_points, size_t m, size_t k, void *out_ext,
                                        void *out_affine) {
    if (!ctx || k == 0 || k > 65535 || (m && !ext_points) || (!out_ext && !out_affine)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "points_sum");
    k_points_sum<<<(unsigned)k, 64, 0, ctx->stream>>>((const uint32_t *)ext_points, m, k, (uint32_t *)out_ext,
                                                     (uint32_t *)out_affine);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_points_validate_dev(vmpc_ctx *ctx, const void *affine, size_t n, uint64_t *n_bad) {
    if (!ctx || !n_bad || (n && !affine)) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_CHECK(vmpc_ws_reserve(ctx, 256));
    unsigned long long *d_bad = (unsigned long long *)vmpc_ws_take(ctx, 8);
    VMPC_HIP_CHECK(hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    if (n) {
        k_points_validate<<<(unsigned)((n + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
            (const uint32_t *)affine, n, d_bad);
        VMPC_KERNEL_CHECK();
    }
    unsigned long long h = 0;
    VMPC_HIP_CHECK(hipMemcpyAsync(&h, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *n_bad = h;
    return VMPC_OK;
}
