// k folds of a tabulated generator vector in one pass.
//
// Protocol 4 halves g_hat every round (verifiable_mpc/ac20/compressed_pivot.py:64: g' = g_l^c * g_r).  After k
// rounds the folded vector is, over the UNFOLDED generators,
//       g^(k)[j] = sum_{b < 2^k} s_b * g[j + b * (n >> k)],        s_b = prod_i (c_i if bit (k - i) of b is 0),
// 2^k scalars shared by all n >> k outputs.  Folding round by round costs a 253-bit scalar multiplication per
// generator and round; with the generators' fixed-base table (msm.hip: row r holds 2^(256 r / R) * g[i]) the k
// rounds collapse into one pass of 64 mixed additions per generator:
//   - the s_b are cut into 64 signed 4-bit digits; digit position p uses row p / O at offset o = p % O, O = 64 / R;
//   - for one offset the R * 2^k table entries of an output form a tiny bucket problem with 8 buckets, the SAME
//     for every output.  The host sorts it once by |digit|, descending; a lane then walks that list with the
//     running-sum form of the bucket reduction (run += entry; at every step down in |digit|: acc += run) - two
//     accumulators, no bucket storage, no divergence (the schedule is wave-uniform);
//   - lane (j, o) writes X[j][o]; a second kernel recombines g^(k)[j] = sum_o 16^o X[j][o] (Horner, 4 doublings a
//     step) and the batched inversion of exact.hip makes the outputs affine.
// A wave holds 64 consecutive outputs at one offset, so every table access is a run of 64 consecutive 128-byte
// lines; the O lanes of an output visit the same R * 2^k lines in different orders, and the workgroups that share
// a range of outputs are placed on the same XCD (blockIdx % 8) so that the re-reads meet in its L2.
// Cost at n = 2^20, R = 4, k = 5: 2^19 lanes x 140 additions.  Used by the prover's round context (prover.hip).
#include <vector>

#include "common.h"
#include "fr.h"
#include "ptio.h"

#define FJ_BLOCK 256
#define FJ_WAVES (FJ_BLOCK / 64)

int vmpc_normalize_launch(vmpc_ctx *ctx, const void *proj, size_t n, void *out_affine);   // exact.hip

// schedule entry: bits 0..7 = b, 8..12 = row, 15 = negate, 16..19 = |digit|; sched[o * e1] = #entries of offset o,
// followed by the entries and a closing entry with |digit| = 0.
// Registers: the running sum and one table entry live in VGPRs (as in k_msm_bucket: 4 waves per SIMD); the
// outer accumulator is touched only at the <= 8 steps down in |digit|, so it lives in its output slot in memory.
// (A first version with both accumulators in registers and a prefetched entry needed 255 VGPRs + 68 spilled.)
__global__ void __launch_bounds__(FJ_BLOCK, 4)
k_fold_jump(const uint32_t *__restrict__ table, size_t stride, size_t m_out, int O, int e1,
            const uint32_t *__restrict__ sched, unsigned n_blocks, uint32_t *__restrict__ partial) {
    const unsigned per_xcd = (n_blocks + 7) / 8;
    const unsigned l = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;      // consecutive l on one XCD
    if (l >= n_blocks) return;
    const unsigned og = (unsigned)O / FJ_WAVES;
    const int o = __builtin_amdgcn_readfirstlane((int)((l % og) * FJ_WAVES + (threadIdx.x >> 6)));
    const size_t j = (size_t)(l / og) * 64 + (threadIdx.x & 63);
    if (j >= m_out) return;                                               // whole waves only when m_out < 64
    const uint32_t *sc = sched + (size_t)o * e1;
    const uint32_t cnt = sc[0];
    const uint32_t *col = table + NIELS_WORDS * j;
    uint32_t *slot = partial + EXT_WORDS * (j * (size_t)O + o);
    ge_ext run = ge_ext_identity();
    ext_st(slot, run);
    bool have = false;
    uint32_t cur = 8;
    for (uint32_t e = 0; e <= cnt; e++) {
        const uint32_t ent = sc[1 + e];
        const uint32_t v = ent >> 16;
        if (cur > v) {                  // entries of |digit| >= cur are all in `run`: it counts once per level
            if (have)
                for (; cur > v; cur--) ext_st(slot, ge_add(ext_ld(slot), run));
            cur = v;
        }
        if (e == cnt) break;
        const ge_niels q = niels_ld_line(col + NIELS_WORDS * (((ent >> 8) & 0x1f) * stride + (size_t)(ent & 0xff) * m_out));
        run = ge_madd(run, ge_niels_select_neg(q, ((ent >> 15) & 1) != 0));
        have = true;
    }
}

__global__ void __launch_bounds__(FJ_BLOCK)
k_fold_jump_combine(const uint32_t *__restrict__ partial, size_t m_out, int O, uint32_t *__restrict__ out_proj) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_out) return;
    const uint32_t *src = partial + EXT_WORDS * j * (size_t)O;
    ge_ext acc = ext_ld(src + EXT_WORDS * (size_t)(O - 1));
    for (int o = O - 2; o >= 0; o--) {
        for (int d = 0; d < 4; d++) acc = ge_dbl(acc);
        acc = ge_add(acc, ext_ld(src + EXT_WORDS * (size_t)o));
    }
    fe_st8(out_proj + 24 * j, acc.X);
    fe_st8(out_proj + 24 * j + 8, acc.Y);
    fe_st8(out_proj + 24 * j + 16, acc.Z);
}

extern "C" int vmpc_msm_table_fold_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                       size_t n_cols, int k, const uint8_t *scalars /* 2^k x 32, host */,
                                       void *out_affine) {
    if (!ctx || !table || !scalars || !out_affine || k < 1 || k > 6 || n_cols < ((size_t)1 << k) ||
        (n_cols & (n_cols - 1)) || n_cols > table_n + table_extra ||
        !(rows == 1 || rows == 2 || rows == 4 || rows == 8 || rows == 16))
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t stride = (table_n + table_extra + 7) & ~(size_t)7;       // msm.hip: msm_table_stride
    const size_t m_out = n_cols >> k;
    const int B = 1 << k, O = 64 / rows, e1 = rows * B + 2;
    // the shared schedule: signed 4-bit digits of the 2^k scalars, per offset sorted by |digit| descending
    std::vector<uint32_t> sched((size_t)O * e1, 0);
    {
        std::vector<std::vector<uint32_t>> by_value((size_t)O * 9);
        for (int b = 0; b < B; b++) {
            uint32_t w[8];
            memcpy(w, scalars + 32 * b, 32);
            if (fr_geq_l(w)) return VMPC_E_NONCANON;
            uint32_t carry = 0;
            for (int p = 0; p < 64; p++) {
                uint32_t raw = ((w[p / 8] >> (4 * (p % 8))) & 15u) + carry;
                int d = (int)raw;
                carry = 0;
                if (raw >= 8) {
                    d = (int)raw - 16;
                    carry = 1;
                }
                if (d == 0) continue;
                const uint32_t v = (uint32_t)(d < 0 ? -d : d);
                by_value[(size_t)(p % O) * 9 + v].push_back((uint32_t)b | ((uint32_t)(p / O) << 8) |
                                                            ((d < 0 ? 1u : 0u) << 15) | (v << 16));
            }
        }
        for (int o = 0; o < O; o++) {
            uint32_t *dst = sched.data() + (size_t)o * e1;
            uint32_t n = 0;
            for (int v = 8; v >= 1; v--)
                for (uint32_t ent : by_value[(size_t)o * 9 + v]) dst[1 + n++] = ent;
            dst[0] = n;
        }
    }
    const size_t sched_bytes = (sched.size() * 4 + 255) & ~(size_t)255;
    const size_t partial_bytes = m_out * (size_t)O * EXT_WORDS * 4;
    const size_t proj_bytes = (m_out * 96 + 255) & ~(size_t)255;
    VMPC_CHECK(vmpc_ws_reserve(ctx, sched_bytes + partial_bytes + proj_bytes));
    char *ws = (char *)ctx->ws;
    uint32_t *d_sched = (uint32_t *)ws, *d_proj = (uint32_t *)(ws + sched_bytes);
    uint32_t *d_partial = (uint32_t *)(ws + sched_bytes + proj_bytes);
    vmpc_stage_scope s(ctx, "table_fold");
    VMPC_CHECK(vmpc_stage_h2d(ctx, d_sched, sched.data(), sched.size() * 4));
    const unsigned n_blocks = (unsigned)(((m_out + 63) / 64) * (size_t)(O / FJ_WAVES));
    const unsigned grid = ((n_blocks + 7) / 8) * 8;
    k_fold_jump<<<grid, FJ_BLOCK, 0, ctx->stream>>>((const uint32_t *)table, stride, m_out, O, e1, d_sched, n_blocks,
                                                    d_partial);
    VMPC_KERNEL_CHECK();
    k_fold_jump_combine<<<(unsigned)((m_out + FJ_BLOCK - 1) / FJ_BLOCK), FJ_BLOCK, 0, ctx->stream>>>(d_partial, m_out, O,
                                                                                                 d_proj);
    VMPC_KERNEL_CHECK();
    return vmpc_normalize_launch(ctx, d_proj, m_out, out_affine);
}
