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

#define RT_THREADS 1024
#define RT_LEAVES 256

__device__ __forceinline__ fe rt_ld(const uint32_t *arr, int idx, int q) { return fe_ld(arr + EXT_WORDS * (size_t)idx + FE_LIMBS * q); }
__device__ __forceinline__ void rt_st(uint32_t *arr, int idx, int q, const fe &v) { fe_st(arr + EXT_WORDS * (size_t)idx + FE_LIMBS * q, v); }
__device__ __forceinline__ fe rt_identity(int q) { return (q == 1 || q == 2) ? fe_one() : fe_zero(); }   // (0 : 1 : 1 : 0)
__device__ __forceinline__ fe rt_add(const fe &P, const fe &Q, int q) { return quadD_add_cached(P, quadD_to_cached(Q, q), q); }
__device__ __forceinline__ int rt_pad16(int v) { return (v + 15) & ~15; }

// The weighted tree over n leaves (a power of two >= 2) held in LDS, in place.  In: TA[t] = A_t and, when present,
// US[t], EX[t] (plain sums that ride along).  Out: RR[0] = sum t A_t, DD[0] = n sum A_t, US[0], EX[0] and, with
// KEEP_T, TA[0] = sum A_t.
// A round covers a contiguous range of j with every kind of job; the quads are laid out in SEGMENTS of one
// operation each (plain additions | R: two additions | D: addition + doubling), every segment starting on a wave,
// so no wave diverges.  A job reads entries 2j, 2j+1 and writes entry j of its arrays: everything a round reads is
// loaded before the barrier that precedes its stores, and earlier rounds only wrote entries below the range - in
// place is safe.
template <bool KEEP_T, bool HAS_U, bool HAS_E>
__device__ __forceinline__ void rt_tree(uint32_t *TA, uint32_t *US, uint32_t *EX, uint32_t *RR, uint32_t *DD, int n) {
    enum { OP_ADD = 0, OP_R = 1, OP_D = 2, OP_PAIR = 3 };
    const int qd = threadIdx.x >> 2, q = threadIdx.x & 3, nquads = blockDim.x >> 2;
    int lvl = 0;
    for (int m = n >> 1; m >= 1; m >>= 1, lvl++) {
        const int nplain = (lvl == 0 ? 0 : (KEEP_T ? 1 : 0)) + (HAS_U ? 1 : 0) + (HAS_E ? 1 : 0);
        int cnt = m;
        while (rt_pad16(cnt * nplain) + (lvl == 0 ? 1 : 2) * rt_pad16(cnt) > nquads) cnt >>= 1;
        const int nA = rt_pad16(cnt * nplain), nC = rt_pad16(cnt);
        const int total = nA + (lvl == 0 ? 1 : 2) * nC;
        // this quad's job within a round
        int op, jl;
        bool on;
        uint32_t *arr;
        {
            int plain_at = lvl == 0 ? nC : 0;
            if (qd >= plain_at && qd < plain_at + nA) {
                int pk = (qd - plain_at) / cnt;
                jl = (qd - plain_at) % cnt;
                on = pk < nplain;
                op = OP_ADD;
                arr = US;
                if (KEEP_T && lvl > 0) {
                    if (pk == 0) arr = TA;
                    pk--;
                }
                if (HAS_U) {
                    if (pk == 0) arr = US;
                    pk--;
                }
                if (HAS_E && pk == 0) arr = EX;
            } else if (lvl == 0) {
                op = OP_PAIR; jl = qd; on = qd < cnt; arr = TA;
            } else if (qd < nA + nC) {
                op = OP_R; jl = qd - nA; on = jl < cnt; arr = RR;
            } else {
                op = OP_D; jl = qd - nA - nC; on = jl < cnt; arr = DD;
            }
            if (!on) arr = TA;
        }
        const bool wave_on = (qd & ~15) < total;
        for (int j0 = 0; j0 < m; j0 += cnt) {
            const int j = j0 + jl, jj = on ? j : 0;
            fe x, y, z;
            if (wave_on) {
                x = rt_ld(arr, 2 * jj, q);
                y = rt_ld(arr, 2 * jj + 1, q);
                if (op == OP_R) z = rt_ld(DD, 2 * jj + 1, q);
            }
            __syncthreads();
            if (wave_on) {
                const fe s = rt_add(x, y, q);
                fe s2 = s;
                if (op == OP_R) s2 = rt_add(s, z, q);
                else if (op == OP_D || op == OP_PAIR) s2 = quadD_dbl(s, q);
                if (on) {
                    if (op == OP_PAIR) {
                        if (KEEP_T) rt_st(TA, j, q, s);
                        rt_st(RR, j, q, y);          // R = 0 * A_2j + 1 * A_2j+1
                        rt_st(DD, j, q, s2);         // D = 2 (A_2j + A_2j+1)
                    } else {
                        rt_st(arr, j, q, s2);
                    }
                }
            }
            __syncthreads();
        }
    }
}

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
__global__ void __launch_bounds__(RT_THREADS)
k_msm_reduce_combine(const uint32_t *__restrict__ in3, int G, int log2L, uint32_t *__restrict__ out,
                     uint32_t *__restrict__ out_packed, uint32_t *done_counter, uint32_t *done_flag, uint32_t done_seq) {
    extern __shared__ __align__(16) uint32_t rt_lds[];
    uint32_t *TA = rt_lds;
    uint32_t *US = TA + (size_t)G * EXT_WORDS;
    uint32_t *EX = US + (size_t)G * EXT_WORDS;
    uint32_t *RR = EX + (size_t)G * EXT_WORDS;
    uint32_t *DD = RR + (size_t)(G > 1 ? G / 2 : 1) * EXT_WORDS;
    const int w = blockIdx.x;
    const uint32_t *src = in3 + (size_t)EXT_WORDS * 3 * (size_t)w * G;
    for (int i = threadIdx.x; i < G * 3 * EXT_WORDS; i += blockDim.x) {
        const int g = i / (3 * EXT_WORDS), r = i % (3 * EXT_WORDS), k = r / EXT_WORDS, e = r % EXT_WORDS;
        uint32_t *dst = k == 0 ? US : k == 1 ? EX : TA;
        dst[g * EXT_WORDS + e] = src[i];
    }
    __syncthreads();
    if (G > 1) rt_tree<false, true, true>(TA, US, EX, RR, DD, G);
    if (threadIdx.x >= 64) return;
    const int q = threadIdx.x & 3;
    fe R = rt_ld(EX, 0, q);
    if (G > 1) R = rt_add(R, rt_ld(RR, 0, q), q);
    for (int k = 0; k < log2L; k++) R = quadD_dbl(R, q);
    R = rt_add(R, rt_ld(US, 0, q), q);
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
    return G >= 1 && G <= 128 && (G & (G - 1)) == 0 && p.chunk_len <= 8 && (p.chunk_len & (p.chunk_len - 1)) == 0;
}

// buckets -> W window sums at w.partials (one per window: k_msm_final's red_blocks = 1); the triples sit behind them.
// out_packed != NULL (one bucket set per commitment, extended output wanted): the window sums ARE the results and go
// there in the public 128-byte form - no recombination launch.
int msm_reduce_tree(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, hipStream_t st, void *out_packed) {
    const int G = p.chunks / RT_LEAVES, L = p.chunk_len;
    const size_t lds_a1 = (size_t)(RT_LEAVES * 2) * EXT_WORDS * 4, lds_aU = (size_t)(RT_LEAVES * 3) * EXT_WORDS * 4;
    const size_t lds_b = (size_t)(4 * G + 1) * EXT_WORDS * 4;
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
    uint32_t *triples = w.partials + (size_t)EXT_WORDS * p.W;
    if (L > 2)
        k_msm_reduce_tree<true, true><<<dim3(G, p.W), RT_THREADS / 2, lds_aU, st>>>(w.buckets, w.counts, p.nb, G, L, triples);
    else if (L > 1)
        k_msm_reduce_tree<true, false><<<dim3(G, p.W), RT_THREADS, lds_aU, st>>>(w.buckets, w.counts, p.nb, G, L, triples);
    else
        k_msm_reduce_tree<false, false><<<dim3(G, p.W), RT_THREADS, lds_a1, st>>>(w.buckets, w.counts, p.nb, G, L, triples);
    VMPC_KERNEL_CHECK();
    k_msm_reduce_combine<<<p.W, RT_THREADS, lds_b, st>>>(triples, G, msm_ilog2(L), w.partials, (uint32_t *)out_packed,
                                                         ctx->d_status + VMPC_ST_WORDS, out_packed ? ctx->done_flag_dev : nullptr,
                                                         ctx->done_seq);
    if (out_packed) ctx->done_flag_dev = nullptr;
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}
