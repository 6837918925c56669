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
#include "quad.h"

#define FJ_BLOCK 256
#define FJ_WAVES (FJ_BLOCK / 64)

int vmpc_normalize_launch(vmpc_ctx *ctx, const void *proj, size_t n, void *out_affine);   // exact.hip

// schedule entry: bits 0..7 = b, 8..12 = row, 15 = negate, 16..19 = |digit|; sched[o * e1] = #entries of offset o,
// followed by the entries and a closing entry with |digit| = 0.
// Registers: the running sum and one table entry live in VGPRs (as in k_msm_bucket: 4 waves per SIMD); the
// outer accumulator is touched only at the <= 8 steps down in |digit|, so it lives in its output slot in memory.
// (A first version with both accumulators in registers and a prefetched entry needed 255 VGPRs + 68 spilled.)
// S > 1 (short vectors): the schedule of an (output, offset) is dealt out to S lanes - workgroup blockIdx.y takes the
// entries e = blockIdx.y mod S of the sorted list - and every lane reduces its share with the same running-sum form;
// the bucket reduction is linear in the set of entries, so the S results simply add up (k_fold_jump_table /
// _combine sum the S slots of an offset).  With 2^11 outputs the pass is 8192 lanes walking 256 table entries each;
// eight lanes walking 32 took the second fold of a proof from 1.2 to 0.5 ms.
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
    const uint32_t S = gridDim.y, share = blockIdx.y;
    const uint32_t *sc = sched + (size_t)o * e1;
    const uint32_t cnt = sc[0];
    const uint32_t *col = table + NIELS_WORDS * j;
    const size_t slot_i = (j * (size_t)O + o) * S + share;
    uint32_t *slot = partial + EXT_WORDS * slot_i;
    uint32_t *park = partial + EXT_WORDS * (m_out * (size_t)O * S + slot_i);   // where `run` waits during a step down
    ge_ext run = ge_ext_identity();
    ext_st(slot, run);
    bool have = false;
    uint32_t cur = 8;
    for (uint32_t e = 0; e <= cnt; e++) {
        const uint32_t ent = sc[1 + e];
        const uint32_t v = ent >> 16;
        if (cur > v) {                  // entries of |digit| >= cur are all in `run`: it counts once per level
            if (have) {
                // the addition of two extended points does not fit beside a live `run` and a table entry in 128
                // registers (193 spilled, the loop below included): `run` goes to memory by hand, here only
                ext_st(park, run);
                asm volatile("" ::: "memory");
                for (; cur > v; cur--) ext_st(slot, ge_add(ext_ld(slot), ext_ld(park)));
                asm volatile("" ::: "memory");
                run = ext_ld(park);
            }
            cur = v;
        }
        if (e == cnt) break;
        if (S > 1 && e % S != share) continue;                            // (wave-uniform)
        const ge_niels q = niels_ld_line(col + NIELS_WORDS * (((ent >> 8) & 0x1f) * stride + (size_t)(ent & 0xff) * m_out));
        run = ge_madd(run, ge_niels_select_neg(q, ((ent >> 15) & 1) != 0));
        have = true;
    }
}

// The same walk with signed 8-BIT digits (round 6): a row of 256 / rows bits then holds 32 / rows offsets instead of
// 64 / rows - HALF the lanes, half the table-line reads (the pass is bound by them: 64 x 128 bytes per generator were
// 8.6 GB at 2^20) - at the price of up to 128 steps down per lane instead of 8.  With half the lanes the chip is full at
// two waves per SIMD, so a lane may hold 256 registers: both accumulators stay in registers and a step down is a plain
// addition.  Schedule entry: bits 0..7 = b, 8..12 = row, 15 = negate, 16..23 = |digit| (1..128).
__global__ void __launch_bounds__(FJ_BLOCK, 2)
k_fold_jump8(const uint32_t *__restrict__ table, size_t stride, size_t m_out, int O, int e1,
             const uint32_t *__restrict__ sched, unsigned n_blocks, uint32_t *__restrict__ partial) {
    const unsigned per_xcd = (n_blocks + 7) / 8;
    const unsigned l = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;      // consecutive l on one XCD
    if (l >= n_blocks) return;
    const unsigned og = (unsigned)O / FJ_WAVES;
    const int o = __builtin_amdgcn_readfirstlane((int)((l % og) * FJ_WAVES + (threadIdx.x >> 6)));
    const size_t j = (size_t)(l / og) * 64 + (threadIdx.x & 63);
    if (j >= m_out) return;                                               // whole waves only when m_out < 64
    const uint32_t S = gridDim.y, share = blockIdx.y;
    const uint32_t *sc = sched + (size_t)o * e1;
    const uint32_t cnt = sc[0];
    const uint32_t *col = table + NIELS_WORDS * j;
    uint32_t *slot = partial + EXT_WORDS * ((j * (size_t)O + o) * S + share);
    ge_ext run = ge_ext_identity(), acc = ge_ext_identity();
    bool have = false;
    uint32_t cur = 128;
    for (uint32_t e = 0; e <= cnt; e++) {
        const uint32_t ent = sc[1 + e];
        const uint32_t v = (ent >> 16) & 0xffu;
        if (cur > v) {                  // entries of |digit| >= cur are all in `run`: it counts once per level
            if (have)
                for (; cur > v; cur--) acc = ge_add(acc, run);
            cur = v;
        }
        if (e == cnt) break;
        if (S > 1 && e % S != share) continue;                            // (wave-uniform)
        const ge_niels q = niels_ld_line(col + NIELS_WORDS * (((ent >> 8) & 0x1f) * stride + (size_t)(ent & 0xff) * m_out));
        run = ge_madd(run, ge_niels_select_neg(q, ((ent >> 15) & 1) != 0));
        have = true;
    }
    ext_st(slot, acc);
}

__global__ void __launch_bounds__(FJ_BLOCK)
k_fold_jump_combine(const uint32_t *__restrict__ partial, size_t m_out, int O, int S, int digit_bits,
                    uint32_t *__restrict__ out_proj) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m_out) return;
    const uint32_t *src = partial + EXT_WORDS * j * (size_t)O * S;
    ge_ext acc = ge_ext_identity();
    for (int o = O - 1; o >= 0; o--) {
        if (o != O - 1)
            for (int d = 0; d < digit_bits; d++) acc = ge_dbl(acc);
        for (int sh = 0; sh < S; sh++) acc = ge_add(acc, ext_ld(src + EXT_WORDS * ((size_t)o * S + sh)));
    }
    fe_st8(out_proj + 24 * j, acc.X);
    fe_st8(out_proj + 24 * j + 8, acc.Y);
    fe_st8(out_proj + 24 * j + 16, acc.Z);
}

// The same recombination for the prover's round context, which wants the folded vector's fixed-base TABLE
// (msm.hip: row r = 2^(256 r / rows) * g', affine, niels form) and nothing else: Horner, the rows' doublings and
// the normalisation in one kernel, a QUAD of lanes per output (quad.h: two dependent multiplications per point
// operation instead of eight).  At 2^15 outputs these are 35 + 240 point operations in a row on an almost empty
// chip - per-lane they took 0.11 + 0.12 (normalise) + 0.66 ms (k_msm_table_build), the time of two rounds.
// One inversion per output covers all its rows, row 0 included (Montgomery's trick, as in k_msm_table_build):
// pass 1 parks (X, Y, Z, Z_0 ... Z_r) in the row's own 128-byte slot, pass 2 (lane 0 of the quad) walks back.
__global__ void __launch_bounds__(FJ_BLOCK)
k_fold_jump_table(const uint32_t *__restrict__ partial, size_t m_out, int O, int S, int digit_bits, size_t stride, int rows,
                  uint32_t *__restrict__ table) {
    const size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    const int q = threadIdx.x & 3;
    if (j >= m_out) return;                                   // whole quads
    const uint32_t *src = partial + EXT_WORDS * j * (size_t)O * S;
    fe P = fe_ld(src + EXT_WORDS * ((size_t)(O - 1) * S) + FE_LIMBS * q);
    for (int sh = 1; sh < S; sh++)
        P = quadD_add_cached(P, quadD_to_cached(fe_ld(src + EXT_WORDS * ((size_t)(O - 1) * S + sh) + FE_LIMBS * q), q), q);
    for (int o = O - 2; o >= 0; o--) {
        for (int d = 0; d < digit_bits; d++) P = quadD_dbl(P, q);
        for (int sh = 0; sh < S; sh++)
            P = quadD_add_cached(P, quadD_to_cached(fe_ld(src + EXT_WORDS * ((size_t)o * S + sh) + FE_LIMBS * q), q), q);
    }
    const int dbl_per_row = 256 / rows;
    fe run = quad_perm<0xaa>(P);                              // Z_0 on every lane
    {
        uint32_t *slot = table + NIELS_WORDS * j;
        fe_st8(slot + 8 * q, quad_sel(P, run, q == 3));
    }
    for (int r = 1; r < rows; r++) {
        for (int d = 0; d < dbl_per_row; d++) P = quadD_dbl(P, q);
        run = fe_mul(run, quad_perm<0xaa>(P));
        uint32_t *slot = table + NIELS_WORDS * ((size_t)r * stride + j);
        fe_st8(slot + 8 * q, quad_sel(P, run, q == 3));
    }
    __threadfence_block();                                    // lane 0 reads what its three neighbours stored
    if (q != 0) return;
    fe inv = fe_inv(run);
    for (int r = rows - 1; r >= 0; r--) {
        uint32_t *slot = table + NIELS_WORDS * ((size_t)r * stride + j);
        const fe Z = fe_ld8(slot + 16);
        const fe prev = r > 0 ? fe_ld8(table + NIELS_WORDS * ((size_t)(r - 1) * stride + j) + 24) : fe_one();
        const fe zi = fe_mul(inv, prev);
        inv = fe_mul(inv, Z);
        ge_aff b;
        b.x = fe_canon(fe_mul(fe_ld8(slot), zi));
        b.y = fe_canon(fe_mul(fe_ld8(slot + 8), zi));
        niels_st_line(slot, ge_niels_from_affine(b));
    }
}

int vmpc_msm_table_build_extras(vmpc_ctx *ctx, size_t n, const void *extra_affine_points, size_t n_extra, int rows,
                                void *table);   // msm.hip

// row blockIdx.y of a narrow table (row_vecs 16-byte vectors per row) into the same row of a wide one
__global__ void k_copy_columns(const uint4 *__restrict__ src, size_t row_vecs, uint4 *__restrict__ dst,
                               size_t dst_row_vecs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < row_vecs) dst[(size_t)blockIdx.y * dst_row_vecs + i] = src[(size_t)blockIdx.y * row_vecs + i];
}

// out_affine != NULL: the folded vector, affine.  Otherwise: its fixed-base table of `out_rows` rows with the
// `out_extra` points (device, affine) as extras, written to `out_table`.
static int table_fold(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows, size_t n_cols,
                      int k, const uint8_t *scalars, void *out_affine, void *out_table, int out_rows,
                      const void *out_extra, size_t out_n_extra, const void *extras_block = nullptr) {
    if (!ctx || !table || !scalars || (!out_affine && !out_table) || k < 1 || k > 6 || n_cols < ((size_t)1 << k) ||
        (n_cols & (n_cols - 1)) || n_cols > table_n + table_extra ||
        !(rows == 1 || rows == 2 || rows == 4 || rows == 8 || rows == 16))
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t stride = (table_n + table_extra + 7) & ~(size_t)7;       // msm.hip: msm_table_stride
    const size_t m_out = n_cols >> k;
    // digit width: 8 bits where a row still holds at least one offset per wave of a workgroup (rows <= 8: the CRS
    // table of a 2^20-generator proof) - half the lanes and table reads of the 4-bit form (k_fold_jump8); else 4
    int digit_bits = (rows <= 8 && ctx->fold_jump_digit_bits != 4) ? 8 : 4;
    const int n_digits = 256 / digit_bits, half = 1 << (digit_bits - 1), per_word = 32 / digit_bits;
    const int B = 1 << k, O = n_digits / rows, e1 = rows * B + 2;
    // the shared schedule: signed digits of the 2^k scalars, per offset sorted by |digit| descending
    std::vector<uint32_t> sched((size_t)O * e1, 0);
    {
        std::vector<std::vector<uint32_t>> by_value((size_t)O * (half + 1));
        for (int b = 0; b < B; b++) {
            uint32_t w[8];
            memcpy(w, scalars + 32 * b, 32);
            if (fr_geq_l(w)) return VMPC_E_NONCANON;
            uint32_t carry = 0;
            for (int p = 0; p < n_digits; p++) {
                uint32_t raw = ((w[p / per_word] >> (digit_bits * (p % per_word))) & (uint32_t)(2 * half - 1)) + carry;
                int d = (int)raw;
                carry = 0;
                if (raw >= (uint32_t)half) {
                    d = (int)raw - 2 * half;
                    carry = 1;
                }
                if (d == 0) continue;
                const uint32_t v = (uint32_t)(d < 0 ? -d : d);
                by_value[(size_t)(p % O) * (half + 1) + v].push_back((uint32_t)b | ((uint32_t)(p / O) << 8) |
                                                                     ((d < 0 ? 1u : 0u) << 15) | (v << 16));
            }
        }
        for (int o = 0; o < O; o++) {
            uint32_t *dst = sched.data() + (size_t)o * e1;
            uint32_t n = 0;
            for (int v = half; v >= 1; v--)
                for (uint32_t ent : by_value[(size_t)o * (half + 1) + v]) dst[1 + n++] = ent;
            dst[0] = n;
        }
    }
    const size_t sched_bytes = (sched.size() * 4 + 255) & ~(size_t)255;
    // short vectors: S lanes share an (output, offset) schedule so that up to 2^17 lanes are at work
    int S = 1;
    // (2^16 / 2^17 / 2^18 lanes: 4.03 / 4.04 / 4.02 ms for the fold of a 2^20-generator proof, 2^19: 4.40 - the pass
    // is bound by its 8 GB of table reads, 64 x 128 bytes per generator, not by the lanes at work)
    while (S < 16 && m_out * (size_t)O * S * 2 <= ((size_t)1 << 17) && rows * B / (S * 2) >= 8) S *= 2;
    const size_t partial_bytes = 2 * m_out * (size_t)O * S * EXT_WORDS * 4;      // slots, then k_fold_jump's parking
    const size_t proj_bytes = (m_out * 96 + 255) & ~(size_t)255;
    VMPC_CHECK(vmpc_ws_reserve(ctx, sched_bytes + partial_bytes + proj_bytes));
    char *ws = (char *)ctx->ws;
    uint32_t *d_sched = (uint32_t *)ws, *d_proj = (uint32_t *)(ws + sched_bytes);
    uint32_t *d_partial = (uint32_t *)(ws + sched_bytes + proj_bytes);
    vmpc_stage_scope s(ctx, "table_fold");
    VMPC_CHECK(vmpc_stage_h2d(ctx, d_sched, sched.data(), sched.size() * 4));
    const unsigned n_blocks = (unsigned)(((m_out + 63) / 64) * (size_t)(O / FJ_WAVES));
    const unsigned grid = ((n_blocks + 7) / 8) * 8;
    if (digit_bits == 8)
        k_fold_jump8<<<dim3(grid, (unsigned)S), FJ_BLOCK, 0, ctx->stream>>>((const uint32_t *)table, stride, m_out, O, e1,
                                                                            d_sched, n_blocks, d_partial);
    else
        k_fold_jump<<<dim3(grid, (unsigned)S), FJ_BLOCK, 0, ctx->stream>>>((const uint32_t *)table, stride, m_out, O, e1,
                                                                           d_sched, n_blocks, d_partial);
    VMPC_KERNEL_CHECK();
    if (out_affine) {
        k_fold_jump_combine<<<(unsigned)((m_out + FJ_BLOCK - 1) / FJ_BLOCK), FJ_BLOCK, 0, ctx->stream>>>(d_partial, m_out,
                                                                                                     O, S, digit_bits, d_proj);
        VMPC_KERNEL_CHECK();
        return vmpc_normalize_launch(ctx, d_proj, m_out, out_affine);
    }
    const size_t out_stride = (m_out + out_n_extra + 7) & ~(size_t)7;
    k_fold_jump_table<<<(unsigned)((4 * m_out + FJ_BLOCK - 1) / FJ_BLOCK), FJ_BLOCK, 0, ctx->stream>>>(
        d_partial, m_out, O, S, digit_bits, out_stride, out_rows, (uint32_t *)out_table);
    VMPC_KERNEL_CHECK();
    if (extras_block) {     // the extras' columns as an (out_stride - m_out)-column table of their own: copy, row by row
        const size_t cols = out_stride - m_out;
        k_copy_columns<<<dim3((unsigned)((cols * NIELS_WORDS / 4 + 63) / 64), (unsigned)out_rows), 64, 0, ctx->stream>>>(
            (const uint4 *)extras_block, cols * NIELS_WORDS / 4, (uint4 *)out_table + m_out * NIELS_WORDS / 4,
            out_stride * NIELS_WORDS / 4);
        VMPC_KERNEL_CHECK();
        return VMPC_OK;
    }
    return vmpc_msm_table_build_extras(ctx, m_out, out_extra, out_n_extra, out_rows, out_table);
}

extern "C" int vmpc_msm_table_fold_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                       size_t n_cols, int k, const uint8_t *scalars /* 2^k x 32, host */,
                                       void *out_affine) {
    if (!out_affine) return VMPC_E_INVAL;
    return table_fold(ctx, table, table_n, table_extra, rows, n_cols, k, scalars, out_affine, nullptr, 0, nullptr, 0);
}

// prover.hip: the extras (k) do not change between proofs - their columns come prebuilt (a table over no
// generators and the same extras: vmpc_msm_table_build_dev(ctx, NULL, 0, extras, n_extra, out_rows, block)),
// valid when (n_cols >> k) is a multiple of 8 so that the block's columns line up
int vmpc_table_fold_table_with_block(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                     size_t n_cols, int k, const uint8_t *scalars, size_t n_extra, int out_rows,
                                     const void *extras_block, void *out_table) {
    return table_fold(ctx, table, table_n, table_extra, rows, n_cols, k, scalars, nullptr, out_table, out_rows, nullptr,
                      n_extra, extras_block);
}

// The fold, then the folded vector's own table (rows x (n_cols >> k + n_extra) entries, vmpc_msm_table_bytes) in
// one go - what the prover's round context continues on.  extra_affine_points: device, n_extra x 64 bytes.
extern "C" int vmpc_msm_table_fold_table_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra,
                                             int rows, size_t n_cols, int k, const uint8_t *scalars,
                                             const void *extra_affine_points, size_t n_extra, int out_rows,
                                             void *out_table) {
    if (!out_table || (n_extra && !extra_affine_points) ||
        !(out_rows == 1 || out_rows == 2 || out_rows == 4 || out_rows == 8 || out_rows == 16))
        return VMPC_E_INVAL;
    return table_fold(ctx, table, table_n, table_extra, rows, n_cols, k, scalars, nullptr, out_table, out_rows,
                      extra_affine_points, n_extra);
}
