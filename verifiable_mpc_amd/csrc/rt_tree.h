// The quad WEIGHT TREE shared by the bucket reduction (msm_reduce_tree.hip) and the fused short-commitment kernel
// (msm_short.hip): helpers for lane-distributed points in LDS and the in-place tree itself.
#pragma once
#include "common.h"
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

