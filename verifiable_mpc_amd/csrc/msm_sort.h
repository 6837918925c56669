// Curve-independent half of the Pippenger pipeline (implemented in msm_sort.hip): signed-digit
// recoding, the two-level per-window bucket sort (coarse bins partitioned in LDS, fine buckets
// resolved per bin), bucket offsets, and the length-balanced segment task table.  Shared by the Ed25519 MSM (msm.hip) and the BN-256
// G1/G2 MSMs (bn256.hip); the curve-specific kernels (entry preparation, bucket accumulation,
// bucket reduction, window recombination) live with their curve.
#pragma once
#include "common.h"

#define MSM_MAX_C 16
#define MSM_WIDE_C 20       // the wide window of a 13-row fixed-base table (msm.hip): int32 digit rows, one set of 2^19 buckets
#define MSM_WIDE_ROWS 13
#define MSM_BLOCK 256
#define MSM_SORT_BLOCK 1024
#define MSM_SEG 64          // length bins of the task table; a segment holds MSM_SEG << seg_shift entries
#define MSM_SEG_LOG2 6
#define MSM_FINISH_SERIAL 32
#ifndef MSM_REDUCE_CHUNKS
#define MSM_REDUCE_CHUNKS 4096
#endif

struct msm_plan {
    size_t n_main, n_extra, n_total;
    int scalar_bits;    // scalars are < 2^scalar_bits
    int c, W, nb, nb1;  // nb = 2^(c-1) buckets per window, nb1 = nb + 1 (bucket 0 unused)
    int seg_shift;      // segments of MSM_SEG << seg_shift entries (negative: MSM_SEG >> -seg_shift), msm_seg_len()
    int balanced;       // 1: a bucket's segments have EQUAL length (+-1); 0: full segments plus a remainder
    int LB;             // fine bits: bucket b = coarse << LB | fine, 2^LB <= 512 fine buckets per coarse bin
    int NC;             // coarse bins per window = nb >> LB
    int J;              // chunks of 8192 terms per digit row
    int idx_bits;       // bits of a term index (ceil log2 n_pad)
    int fine_in_entry;  // the 32-bit entry between the two sort passes carries the fine bucket (idx_bits + LB <= 31)
    int period;         // digit rows per commitment: a batch of K commitments has W = K * period rows
    int top_row;        // rows with (row % period) == top_row are a top window alone (-1: none): LB_top fine bits
    uint32_t top_max_b; // largest bucket index a canonical scalar's top digit reaches
    int LB_top;
    size_t n_pad;       // digit row stride: n_total rounded up to 8 (rows are 16-byte aligned, zero padded)
    int chunks;         // chunk-threads per window in the reduce kernel
    int chunk_len;      // buckets per chunk (power of two)
    int red_blocks;     // blocks per window in the reduce kernel
    // wide window (c = MSM_WIDE_C over a MSM_WIDE_ROWS-row table): digits are int32; a digit row is the flattened
    // [table row][column] and every sort chunk lies inside ONE table row (row_stride = chunks_per_row * 8192), so the
    // entries between the sort passes carry the COLUMN (col_bits) beside the fine bucket and the table row is read
    // off the entry's place in its coarse bin (runs are laid out chunk by chunk, i.e. row by row)
    int wide = 0;
    int chunks_per_row = 0; // 0: entries carry the flattened index (every other plan)
    int col_bits = 0;
    size_t row_stride = 0;
    int fine_cap = 12288;   // entries of a coarse bin that k_sort_fine stages in LDS (16384 for the wide window)
};
inline size_t msm_seg_len(const msm_plan &p) { return (size_t)1 << (MSM_SEG_LOG2 + p.seg_shift); }


struct msm_ws {
    uint32_t *entries;      // prepared points, entry_bytes each
    uint32_t *hist1;        // [W][NC][J] coarse-bin counts per chunk, scanned in place; [hist1_n] = #entries
    size_t hist1_n;
    uint32_t *stage1;       // entries partitioned by (window, coarse bin)
    uint32_t *counts, *starts, *sorted;
    uint32_t *buckets;      // W * nb accumulators, acc_bytes each
    uint32_t *partials;     // W * red_blocks accumulators (msm_reduce_tree: W window sums, then 3 W red_blocks)
    uint32_t *nseg, *seg_starts, *block_hist, *block_base, *heavy_list, *ctrl;
    uint32_t *seg_partial;  // per-segment partial sums, acc_bytes each
    uint2 *tasks;
    int16_t *digits;        // int32 rows when the plan is `wide`
    void *scan_ws;
    size_t total;
    uint32_t plan_blocks;
    size_t t_max;
};

struct msm_modulus {
    uint32_t v[8];
};

// `modulus` (scalars are canonical residues below it) lets the plan bound the top window's digits
void msm_make_plan(vmpc_ctx *ctx, size_t n_main, size_t n_extra, int scalar_bits, msm_plan &p,
                   const msm_modulus *modulus = nullptr);
void msm_plan_geometry(vmpc_ctx *ctx, msm_plan &p);   // from (n_total, c, W) already set
void msm_layout(const msm_plan &p, msm_ws &w, char *base, size_t entry_bytes, size_t acc_bytes);
// Ed25519 bucket reduction through the quad tree (msm_reduce_tree.hip): W window sums at w.partials
bool msm_reduce_tree_fits(const msm_plan &p);
int msm_reduce_tree_split(const msm_plan &p);      // partial results per window that k_msm_final adds (1: none)
int msm_reduce_tree(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, hipStream_t st, void *out_packed);
// recode -> hist1 -> scan -> part1 -> fine -> plan: fills digits, sorted, starts, counts, nseg,
// seg_starts, heavy_list, tasks, ctrl[0] = #split buckets, ctrl[1] = #tasks
int msm_sort_stage(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, const void *scalars, size_t n,
                   const void *extra_scalars, const msm_modulus &modulus);
int msm_sort_digits(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w);   // the same minus the recoding
// wide window: the MSM_WIDE_ROWS signed 20-bit digits of the K commitments' scalars (main || extra) as int32 rows,
// digit r of column i of commitment k at digits32[k * digits_per_commitment + r * row_stride + i]
int msm_recode_wide_batch(vmpc_ctx *ctx, const void *const *scalars, size_t n_main, const void *const *extra_scalars,
                          int K, size_t extra_pos, size_t n_extra, size_t row_stride, int32_t *digits32,
                          size_t digits_per_commitment, const msm_modulus &modulus);
// signed c-bit digits of (main || extra) scalars for a fixed-base table of `rows` rows (msm.hip): digit
// w of term i goes to digits[(w % (W/rows)) * rows * n_pad + (w / (W/rows)) * n_pad + i]; positions
// outside [0, n_main) and [extra_pos, extra_pos + n_extra) are zeroed
int msm_recode_rows(vmpc_ctx *ctx, const void *scalars, size_t n_main, const void *extra_scalars,
                    size_t extra_pos, size_t n_extra, size_t n_pad, int16_t *digits, int c, int W, int rows,
                    const msm_modulus &modulus);
// the K commitments of one pass in one launch: commitment k writes its digit rows at digits + k * digits_per_commitment
int msm_recode_rows_batch(vmpc_ctx *ctx, const void *const *scalars, size_t n_main, const void *const *extra_scalars,
                          int K, size_t extra_pos, size_t n_extra, size_t n_pad, int16_t *digits,
                          size_t digits_per_commitment, int c, int W, int rows, const msm_modulus &modulus);

static inline int msm_ilog2(int v) {
    int r = 0;
    while ((1 << r) < v) r++;
    return r;
}

// A bucket of cnt entries is cut into ns = ceil(cnt / seg) segments.  balanced: of EQUAL length (+-1), the first
// cnt % ns of them one entry longer; otherwise full segments of `seg` entries plus a remainder.  Which is better
// depends on how the tasks fill the chip: with ~2 x seg entries per bucket and four waves per SIMD (the A_i / B_i pair
// of an Ed25519 prover round) the remainders formed a second, short wave of tasks on a quarter of the chip - equal
// segments took that bucket stage from 0.75 to 0.71 ms; with one wave per SIMD (the 384-VGPR BN-256 kernels run their
// workgroups in rounds) the short remainders ARE the cheap last round - equal segments cost 14 %.  The plan says.
// Segment sidx of the bucket: its first entry and its length.
__host__ __device__ __forceinline__ uint32_t msm_seg_count(uint32_t cnt, uint32_t seg_log) {
    return (cnt + (1u << seg_log) - 1u) >> seg_log;
}
__device__ __forceinline__ void msm_seg_range(uint32_t cnt, uint32_t ns, uint32_t sidx, uint32_t seg, int balanced,
                                              uint32_t &off, uint32_t &len) {
    if (ns <= 1) {
        off = 0;
        len = cnt;
        return;
    }
    if (!balanced) {
        off = sidx * seg;
        len = cnt - off < seg ? cnt - off : seg;
        return;
    }
    const uint32_t q = cnt / ns, r = cnt - q * ns;
    off = sidx * q + (sidx < r ? sidx : r);
    len = q + (sidx < r ? 1u : 0u);
}

// ci = w * nb1 + b (b >= 1)  ->  w * nb + (b - 1)
__device__ __forceinline__ size_t msm_bucket_slot(uint32_t ci, int nb1) {
    uint32_t w = ci / (uint32_t)nb1;
    return (size_t)ci - w - 1;
}
