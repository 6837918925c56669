// Bucket reduction  R_w = sum_b b * B_b  of one window WITHOUT per-lane offset ladders (latency path).
//
// k_msm_reduce (msm.hip) gives every chunk of L consecutive buckets a lane that multiplies its chunk sum by the
// chunk's offset with a double-and-add ladder: 2^16 lanes x ~33 point operations when every bucket is its own
// chunk (the prover's rounds: one or two bucket sets of 2^15 buckets) - 2.2 M point operations for 2^16 buckets,
// and the stage is bound by that work (~125 us).  Here the weights come out of a TREE:
//   a node over leaves [s, s + 2^k) carries  R = sum (t - s) A_t  and  D = 2^k sum A_t;
//   merging two nodes of 2^k leaves:  R = R_l + R_r + D_r,  D = 2 (D_l + D_r)       (leaves: R = 0, D = A_t)
// - four point operations per merge, ~1000 per 256 leaves instead of 8448, and the depth of the whole reduction
// is 2 operations per level.  Every operation is done by a QUAD of lanes (quad.h: 2-3 dependent field
// multiplications per point operation instead of 8-9), one quad per merge job.  One wave of such jobs keeps its
// SIMD's VALU busy by itself (a field multiplication is ~170 issue-bound instructions), so the jobs of a level are
// packed on as few waves as possible, one kind of job per wave, and waves without a job only meet the barriers.
//
// Geometry: a window has K = 256 G chunks of L buckets.  Workgroup g of the window takes the chunks
// c = 256 g + t (t = 0..255: its leaves).  Its tree's root holds Rw_g = sum_t t A_t and D_g = 256 sum_t A_t -
// the very multiple the tree over the G workgroups of the window (second kernel) needs for its leaves:
//   R_w = sum_c [S_c + c L A_c]            A_c = sum_j B_{cL+j},  S_c = sum_j (j + 1) B_{cL+j}   (bucket value = index + 1)
//       = sum_g U_g + L (sum_g Rw_g + sum_g g D_g)          U_g = sum_t S_{256 g + t}   (L = 1: S = A)
#include "common.h"
#include "msm_sort.h"
#include "ge25519.h"
#include "ptio.h"
#include "quad.h"

#include "rt_tree.h"

// workgroup (g, w): chunks 256 g .. 256 g + 255 of window w; out3[(w G + g) 3 + {0, 1, 2}] = U_g, Rw_g, D_g
// Chunk sums: L <= 2 - a quad per chunk (1024 threads); longer chunks - a LANE per chunk (LANE_P1: the running sums
// are throughput, four waves of lanes do 2 L additions of 9 multiplications where sixteen waves of quads would do
// them at 12), in a workgroup of 512 threads so that the lane form has its 256 registers.
template <bool HAS_U, bool LANE_P1>
__global__ void __launch_bounds__(LANE_P1 ? RT_THREADS / 2 : RT_THREADS)
k_msm_reduce_tree(const uint32_t *__restrict__ buckets, const uint32_t *__restrict__ counts, int nb, int G, int L,
                  uint32_t *__restrict__ out3) {
    extern __shared__ __align__(16) uint32_t rt_lds[];
    uint32_t *TA = rt_lds;
    uint32_t *RR = TA + RT_LEAVES * EXT_WORDS;
    uint32_t *DD = RR + (RT_LEAVES / 2) * EXT_WORDS;
    uint32_t *US = DD + (RT_LEAVES / 2) * EXT_WORDS;             // only with HAS_U
    const int w = blockIdx.y, g = blockIdx.x;
    if (LANE_P1) {
        if (threadIdx.x < RT_LEAVES) {
            const int t = threadIdx.x;
            const int lo = (RT_LEAVES * g + t) * L;
            const uint32_t *bw = buckets + EXT_WORDS * ((size_t)w * nb + lo);
            const uint32_t *cw = counts + (size_t)w * (nb + 1) + lo + 1;
            ge_ext acc = ge_ext_identity(), sum = ge_ext_identity();
            bool have = false;
            for (int j = L - 1; j >= 0; j--) {
                if (cw[j]) {
                    const ge_ext B = ext_ld(bw + EXT_WORDS * j);
                    acc = have ? ge_add(acc, B) : B;
                    have = true;
                }
                if (j == L - 1) sum = acc;
                else if (have) sum = ge_add(sum, acc);
            }
            ext_st(TA + EXT_WORDS * t, acc);
            if (HAS_U) ext_st(US + EXT_WORDS * t, sum);
        }
    } else {
        const int t = threadIdx.x >> 2, q = threadIdx.x & 3;
        const int lo = (RT_LEAVES * g + t) * L;                  // 0-based bucket index; bucket value = index + 1
        const uint32_t *bw = buckets + EXT_WORDS * ((size_t)w * nb + lo);
        const uint32_t *cw = counts + (size_t)w * (nb + 1) + lo + 1;   // empty buckets are never written
        fe acc = rt_identity(q), sum = acc;
        bool have = false;                                       // quad-uniform, like every condition below
        for (int j = L - 1; j >= 0; j--) {
            if (cw[j]) {
                const fe B = fe_ld(bw + EXT_WORDS * j + FE_LIMBS * q);
                acc = have ? rt_add(acc, B, q) : B;
                have = true;
            }
            if (HAS_U) {
                if (j == L - 1) sum = acc;
                else if (have) sum = rt_add(sum, acc, q);
            }
        }
        rt_st(TA, t, q, acc);
        if (HAS_U) rt_st(US, t, q, sum);
    }
    __syncthreads();
    rt_tree<!HAS_U, HAS_U, false>(TA, US, nullptr, RR, DD, RT_LEAVES);
    uint32_t *o = out3 + (size_t)EXT_WORDS * 3 * ((size_t)w * G + g);
    if (threadIdx.x < EXT_WORDS) {
        o[threadIdx.x] = HAS_U ? US[threadIdx.x] : TA[threadIdx.x];
        o[EXT_WORDS + threadIdx.x] = RR[threadIdx.x];
        o[2 * EXT_WORDS + threadIdx.x] = DD[threadIdx.x];
    }
}

// workgroup w: R_w = X + L (Y + Z) from the G triples of the window:  X = sum U_g, Y = sum Rw_g, Z = sum g D_g
// spl > 1 (msm_short.hip): every group's triple arrives as spl PARTIAL triples (the group's entries were shared out
// among spl workgroups); a triple is linear in the bucket sums, so the parts are added on the way in.
// reset: words this workgroup zeroes for the next call (the short path's bin cursors), or NULL.
// sub > 1 (one set of 2^19 buckets: 256 groups of a window, more than one workgroup's LDS holds): workgroup w * sub + h
// takes the groups [h G, (h + 1) G) of window w and adds their offset h G sum_g D_g = h * DD[0] to Z; the `sub` partial
// results of a window are consecutive in `out` and k_msm_final (red_blocks = sub) adds them.
__global__ void __launch_bounds__(RT_THREADS)
k_msm_reduce_combine(const uint32_t *__restrict__ in3, int G, int log2L, uint32_t *__restrict__ out,
                     uint32_t *__restrict__ out_packed, uint32_t *done_counter, uint32_t *done_flag, uint32_t done_seq,
                     int spl, uint32_t *__restrict__ reset, int reset_words, const uint32_t *__restrict__ poison,
                     int sub) {
    extern __shared__ __align__(16) uint32_t rt_lds[];
    uint32_t *TA = rt_lds;
    uint32_t *US = TA + (size_t)G * EXT_WORDS;
    uint32_t *EX = US + (size_t)G * EXT_WORDS;
    uint32_t *RR = EX + (size_t)G * EXT_WORDS;
    uint32_t *DD = RR + (size_t)(G > 1 ? G / 2 : 1) * EXT_WORDS;
    const int w = blockIdx.x;
    const uint32_t *src = in3 + (size_t)EXT_WORDS * 3 * (size_t)w * G * spl;
    if (reset)
        for (int i = threadIdx.x; i < reset_words; i += blockDim.x) reset[(size_t)w * reset_words + i] = 0;
    for (int i = threadIdx.x; i < G * 3 * EXT_WORDS; i += blockDim.x) {
        const int g = i / (3 * EXT_WORDS), r = i % (3 * EXT_WORDS), k = r / EXT_WORDS, e = r % EXT_WORDS;
        uint32_t *dst = k == 0 ? US : k == 1 ? EX : TA;
        dst[g * EXT_WORDS + e] = src[(size_t)g * spl * 3 * EXT_WORDS + r];
    }
    __syncthreads();
    for (int part = 1; part < spl; part++) {
        // one quad per (group, array): LDS entry += the part's entry, read straight from global memory
        const int q = threadIdx.x & 3;
        for (int job = threadIdx.x >> 2; job < 3 * G; job += blockDim.x >> 2) {
            const int g = job / 3, k = job % 3;
            uint32_t *dst = k == 0 ? US : k == 1 ? EX : TA;
            const fe y = fe_ld(src + ((size_t)(g * spl + part) * 3 + k) * EXT_WORDS + FE_LIMBS * q);
            rt_st(dst, g, q, rt_add(rt_ld(dst, g, q), y, q));
        }
        __syncthreads();
    }
    if (G > 1) rt_tree<false, true, true>(TA, US, EX, RR, DD, G);
    if (threadIdx.x >= 64) return;
    const int q = threadIdx.x & 3;
    fe R = rt_ld(EX, 0, q);
    if (G > 1) R = rt_add(R, rt_ld(RR, 0, q), q);
    if (sub > 1 && G > 1) {
        const fe D = rt_ld(DD, 0, q);                         // G * sum of this part's leaves
        for (int h = w % sub; h > 0; h--) R = rt_add(R, D, q);
    }
    for (int k = 0; k < log2L; k++) R = quadD_dbl(R, q);
    R = rt_add(R, rt_ld(US, 0, q), q);
    // poison[w] != 0 (msm_short.hip: THIS commitment's scalars were beyond the fused path's capacities): the result is
    // void and goes out as all zeros - Z = 0 is no point, so a consumer can tell which of several commitments on a
    // context it was, whoever collected the context's status word
    if (poison && poison[w]) R = fe_zero();
    if (threadIdx.x < 4) {
        if (out_packed) fe_st8(out_packed + 32 * (size_t)w + 8 * q, R);      // one bucket set per commitment: its result
        else fe_st(out + EXT_WORDS * (size_t)w + FE_LIMBS * q, R);
    }
    // only wave 0 is left here (no workgroup barrier after the early return above).  The four lanes' stores and lane
    // 0's publish are instructions of ONE wave, issued in program order; the release fence inside vmpc_publish_done
    // waits for all of the wave's outstanding stores (the memory counters are per wave).  The wave barrier keeps the
    // compiler from moving the stores below it.
    __builtin_amdgcn_wave_barrier();
    if (out_packed && done_flag && threadIdx.x == 0) vmpc_publish_done(done_counter, done_flag, done_seq);
}

bool msm_reduce_tree_fits(const msm_plan &p) {
    if (p.chunks % RT_LEAVES != 0 || p.chunks * p.chunk_len != p.nb) return false;
    const int G = p.chunks / RT_LEAVES;
    // (the one set of 2^19 buckets of a wide-window commitment: chunks of 16 buckets, or 256 groups combined in two
    // halves - msm_reduce_tree_split)
    return G >= 1 && (G <= 128 || (p.wide && G == 256)) && (G & (G - 1)) == 0 && p.chunk_len <= (p.wide ? 32 : 8) &&
           (p.chunk_len & (p.chunk_len - 1)) == 0;
}

// parts a window's groups are combined in (k_msm_reduce_combine `sub`): the window's result is the sum of that many
// partial results
int msm_reduce_tree_split(const msm_plan &p) {
    const int G = p.chunks / RT_LEAVES;
    return G > 128 ? G / 128 : 1;
}

// buckets -> W window sums at w.partials (one per window: k_msm_final's red_blocks = 1); the triples sit behind them.
// out_packed != NULL (one bucket set per commitment, extended output wanted): the window sums ARE the results and go
// there in the public 128-byte form - no recombination launch.
int msm_reduce_tree(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, hipStream_t st, void *out_packed) {
    const int G = p.chunks / RT_LEAVES, L = p.chunk_len;
    const size_t lds_a1 = (size_t)(RT_LEAVES * 2) * EXT_WORDS * 4, lds_aU = (size_t)(RT_LEAVES * 3) * EXT_WORDS * 4;
    if (!ctx->reduce_tree_ready) {
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_msm_reduce_tree<false, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a1));
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_msm_reduce_tree<true, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_aU));
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_msm_reduce_tree<true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_aU));
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_msm_reduce_combine,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4 * 128 + 1) * EXT_WORDS * 4)));
        ctx->reduce_tree_ready = true;
    }
    const int sub = msm_reduce_tree_split(p);
    uint32_t *triples = w.partials + (size_t)EXT_WORDS * p.W * sub;
    if (L > 2)
        k_msm_reduce_tree<true, true><<<dim3(G, p.W), RT_THREADS / 2, lds_aU, st>>>(w.buckets, w.counts, p.nb, G, L, triples);
    else if (L > 1)
        k_msm_reduce_tree<true, false><<<dim3(G, p.W), RT_THREADS, lds_aU, st>>>(w.buckets, w.counts, p.nb, G, L, triples);
    else
        k_msm_reduce_tree<false, false><<<dim3(G, p.W), RT_THREADS, lds_a1, st>>>(w.buckets, w.counts, p.nb, G, L, triples);
    VMPC_KERNEL_CHECK();
    if (sub > 1 && out_packed) return VMPC_E_INVAL;            // (the caller lets k_msm_final add the parts)
    k_msm_reduce_combine<<<p.W * sub, RT_THREADS, (size_t)(4 * (G / sub) + 1) * EXT_WORDS * 4, st>>>(
        triples, G / sub, msm_ilog2(L), w.partials, (uint32_t *)out_packed, ctx->d_status + VMPC_ST_WORDS,
        out_packed ? ctx->done_flag_dev : nullptr, ctx->done_seq, 1, nullptr, 0, nullptr, sub);
    if (out_packed) ctx->done_flag_dev = nullptr;
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

// the second kernel alone, for triples somebody else produced (msm_short.hip): W windows of G groups x spl parts
int msm_reduce_combine_launch(vmpc_ctx *ctx, const uint32_t *triples, int W, int G, int spl, uint32_t *scratch_out,
                              void *out_packed, uint32_t *reset, int reset_words, const uint32_t *poison) {
    hipStream_t st = ctx->stream;
    if (!ctx->reduce_tree_ready && !ctx->short_ready)
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_msm_reduce_combine,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4 * 128 + 1) * EXT_WORDS * 4)));
    const size_t lds_b = (size_t)(4 * G + 1) * EXT_WORDS * 4;
    k_msm_reduce_combine<<<W, RT_THREADS, lds_b, st>>>(triples, G, 0, scratch_out, (uint32_t *)out_packed,
                                                       ctx->d_status + VMPC_ST_WORDS, out_packed ? ctx->done_flag_dev : nullptr,
                                                       ctx->done_seq, spl, reset, reset_words, poison, 1);
    if (out_packed) ctx->done_flag_dev = nullptr;
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}
