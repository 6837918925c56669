// Curve-independent half of the Pippenger pipeline for gfx950 (shared by msm.hip and bn256.hip):
//
//   recode   32-byte scalars -> W signed c-bit digits (int16 rows, 16-byte aligned)
//   hist1    per (chunk of 8192 terms, window): LDS histogram of COARSE bins (high bits of the bucket)
//   scan     one exclusive scan over [window][coarse bin][chunk] -> where every chunk's run of a bin goes
//   part1    per (chunk, window): counting sort of the chunk by coarse bin INSIDE LDS, then each bin's
//            run leaves as one contiguous piece (64 entries = 256 B at n = 2^20) - whole lines, not the
//            4-byte scattered stores of a one-pass bucket sort
//   fine     per (coarse bin, window): the bin's entries (contiguous, ~8 K, L2 resident) are read twice:
//            histogram of the <= 512 FINE buckets in LDS, then scattered to their final, bucket-sorted
//            place - a 32-KiB region that this one workgroup fills completely within microseconds, so
//            the 4-byte stores merge in its XCD's L2.  Bucket counts and start offsets fall out here.
//   plan     cut every bucket's run into <= 64-entry segments, number them by length (tasks)
//
// An entry between part1 and fine is  index | fine bucket << idx_bits | sign << 31  (4 bytes): the
// digits array is read sequentially twice and never gathered.  Replaces the round-1 (slice, bucket range)
// one-pass sort, which moved 559 MB for 67 MB of indices (partial-line read-modify-writes) and re-read
// every digit row once per bucket range.
//
// Replaces the O(n) Python loop `[g[i] ** _int(x_i) for i, x_i in enumerate(x)]` of
// verifiable_mpc/ac20/pivot.py:143 only in the sense that it orders its terms for the bucket method; the
// arithmetic lives with the curve.
#include "common.h"
#include "msm_sort.h"
#include "scan.h"

#define SORT_T 8192            // terms per chunk = 1024 threads x one 16-byte vector of 8 digits
#define SORT_BLOCK 1024
#define SORT_WAVES (SORT_BLOCK / 64)
#define SORT_FINE_CAP 12288    // entries of a coarse bin staged in LDS (48 KiB): 1.5 x the 8 K target

// length class of a bucket's last, partial segment: 1 .. MSM_SEG - 1 in units of 2^seg_shift entries (segments of
// more than MSM_SEG entries) or of one entry (seg_shift <= 0: segments of MSM_SEG >> -seg_shift entries)
__device__ __forceinline__ uint32_t msm_seg_class(uint32_t rem, int seg_shift) {
    return seg_shift > 0 ? (rem + (1u << seg_shift) - 1u) >> seg_shift : rem;
}

__device__ __forceinline__ void load_u32x8(uint32_t dst[8], const uint32_t *src) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    uint4 a = p[0], b = p[1];
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
}

// ---- recode: scalar -> signed digits ----------------------------------------------------
// Row w of `digits` holds window w of every term: main terms at [0, n_main), extra terms at
// [extra_pos, extra_pos + n_extra), zeros (= no entry) everywhere else up to the row stride n_pad.
struct msm_recode_batch {          // scalar vectors of the commitments of one pass (vmpc_msm_table_batch_dev)
    const uint32_t *sc[16];
    const uint32_t *sc_extra[16];
};

__device__ __forceinline__ void
msm_recode_term(const uint32_t *__restrict__ sc, size_t n_main, const uint32_t *__restrict__ sc_extra,
                size_t extra_pos, size_t n_extra, size_t n_pad, int16_t *__restrict__ digits, int c, int W,
                int wpr, size_t set_stride, const msm_modulus &mod, uint32_t *__restrict__ status) {
    // digit w of term i goes to digits[(w % wpr) * set_stride + (w / wpr) * n_pad + i]: plain MSMs have
    // wpr = W and set_stride = n_pad (row w = window w); fixed-base tables of r rows have wpr = W / r
    // bucket sets, each a row of r * n_pad entries whose index is the table position (w / wpr, i)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    const uint32_t *src = nullptr;
    if (i < n_main) src = sc + 8 * i;
    else if (sc_extra && i >= extra_pos && i < extra_pos + n_extra) src = sc_extra + 8 * (i - extra_pos);
    if (!src) {                              // the sort kernels read whole 16-byte vectors of a row
        for (int w = 0; w < W; w++) digits[(size_t)(w % wpr) * set_stride + (size_t)(w / wpr) * n_pad + i] = 0;
        return;
    }
    uint32_t s[8];
    load_u32x8(s, src);
    {   // canonical residue? (s < modulus); the caller is told at the next sync point
        bool ge = true;
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            if (s[k] != mod.v[k]) {
                ge = s[k] > mod.v[k];
                break;
            }
        }
        if (ge) {
            // the call fails with VMPC_E_NONCANON at its sync point; until then the term counts as zero, so
            // that every digit the sort sees obeys the bounds the plan derived from the modulus
            atomicAdd(&status[VMPC_ST_NONCANON], 1u);
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = 0;
        }
    }
    const uint32_t mask = (1u << c) - 1u;
    const uint32_t half = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
        uint32_t raw = (s[0] & mask) + carry;
        int32_t d;
        if (raw >= half) {
            d = (int32_t)raw - (int32_t)(1u << c);
            carry = 1;
        } else {
            d = (int32_t)raw;
            carry = 0;
        }
        digits[(size_t)(w % wpr) * set_stride + (size_t)(w / wpr) * n_pad + i] = (int16_t)d;
        // s >>= c  (c < 32; static limb indices keep s[] in registers)
#pragma unroll
        for (int k = 0; k < 7; k++) s[k] = (s[k] >> c) | (s[k + 1] << (32 - c));
        s[7] >>= c;
    }
}

__global__ void __launch_bounds__(MSM_BLOCK)
k_msm_recode(const uint32_t *__restrict__ sc, size_t n_main, const uint32_t *__restrict__ sc_extra,
             size_t extra_pos, size_t n_extra, size_t n_pad, int16_t *__restrict__ digits, int c, int W,
             int wpr, size_t set_stride, msm_modulus mod, uint32_t *__restrict__ status) {
    msm_recode_term(sc, n_main, sc_extra, extra_pos, n_extra, n_pad, digits, c, W, wpr, set_stride, mod, status);
}

// blockIdx.y = commitment of the batch: its own scalar vectors, its own wpr digit rows
__global__ void __launch_bounds__(MSM_BLOCK)
k_msm_recode_batch(msm_recode_batch b, size_t n_main, size_t extra_pos, size_t n_extra, size_t n_pad,
                   int16_t *__restrict__ digits, size_t digits_per_commitment, int c, int W, int wpr,
                   size_t set_stride, msm_modulus mod, uint32_t *__restrict__ status) {
    msm_recode_term(b.sc[blockIdx.y], n_main, b.sc_extra[blockIdx.y], extra_pos, n_extra, n_pad,
                    digits + (size_t)blockIdx.y * digits_per_commitment, c, W, wpr, set_stride, mod, status);
}

// ---- wide window: 13 signed 20-bit digits per scalar, int32 rows --------------------------------
// Signed recoding without a carry chain: with h = 2^19 and H = h * sum_r 2^(20 r), digit r of s is
// ((s + H) >> 20 r & (2^20 - 1)) - h, in [-h, h) - the very digits the carry loop of msm_recode_term produces
// (raw + carry >= h  <=>  the addition of h at position r carries out).  s < l < 2^253 and H < 2^260: thirteen digits,
// the top one in [0, 2^13].
__global__ void __launch_bounds__(MSM_BLOCK)
k_msm_recode_wide(msm_recode_batch b, size_t n_main, size_t extra_pos, size_t n_extra, size_t row_stride,
                  int32_t *__restrict__ digits, size_t digits_per_commitment, msm_modulus mod,
                  uint32_t *__restrict__ status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_stride) return;
    const uint32_t *sc = b.sc[blockIdx.y], *sc_extra = b.sc_extra[blockIdx.y];
    int32_t *out = digits + (size_t)blockIdx.y * digits_per_commitment + i;
    const uint32_t *src = nullptr;
    if (i < n_main) src = sc + 8 * i;
    else if (sc_extra && i >= extra_pos && i < extra_pos + n_extra) src = sc_extra + 8 * (i - extra_pos);
    if (!src) {
#pragma unroll
        for (int r = 0; r < MSM_WIDE_ROWS; r++) out[(size_t)r * row_stride] = 0;
        return;
    }
    uint32_t s[9];
    load_u32x8(s, src);
    s[8] = 0;
    {
        bool ge = true;          // canonical residue?  (as msm_recode_term)
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            if (s[k] != mod.v[k]) {
                ge = s[k] > mod.v[k];
                break;
            }
        }
        if (ge) {
            atomicAdd(&status[VMPC_ST_NONCANON], 1u);
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = 0;
        }
    }
    // The top digit of a canonical scalar has 13 bits: 2^20 entries of table row 12 would crowd into the lowest 2^13 of
    // the 2^19 buckets (16 of the 1024 coarse bins six times as full as the rest: 227 us of sorting instead of ~50).
    // l * P = O for every generator, so s may be replaced by s + k l: with k in [0, 120] (s + k l < 2^259: the top
    // digit stays below 2^19) chosen by a hash of the column, the top digit 4096 k + (s >> 240) is spread over 94 %
    // of the buckets like the other twelve.  Scalars below 2^240 (zeros and the small wire values of a witness, whose
    // few digits are what makes them cheap) are left as they are.
    if (mod.v[7] == 0x10000000u && (s[7] >> 16) != 0) {
        const uint32_t k = ((((uint32_t)i * 2654435761u) >> 16) * 121u) >> 16;
        uint64_t c2 = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            c2 += (uint64_t)s[w] + (uint64_t)k * mod.v[w];
            s[w] = (uint32_t)c2;
            c2 >>= 32;
        }
        s[8] = (uint32_t)c2;
    }
    // H = 2^19 * (1 + 2^20 + ... + 2^240): bit 19 + 20 r set
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint32_t hk = 0;
#pragma unroll
        for (int r = 0; r < MSM_WIDE_ROWS; r++)
            if ((19 + 20 * r) / 32 == k) hk |= 1u << ((19 + 20 * r) % 32);
        const uint64_t v = (uint64_t)s[k] + hk + carry;
        s[k] = (uint32_t)v;
        carry = v >> 32;
    }
#pragma unroll
    for (int r = 0; r < MSM_WIDE_ROWS; r++) {
        const int bit = 20 * r, k = bit / 32, sh = bit % 32;
        uint32_t f = s[k] >> sh;
        if (sh > 12) f |= s[k + 1] << (32 - sh);
        out[(size_t)r * row_stride] = (int32_t)(f & 0xfffffu) - (int32_t)(1u << 19);
    }
}

// eight consecutive digits of a row: one 16-byte vector of int16, or two of int32 (wide window)
template <bool WIDE>
__device__ __forceinline__ void sort_load8(const int16_t *__restrict__ digits, size_t at, int d[8]) {
    if (WIDE) {
        const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const int32_t *>(digits) + at);
        const uint4 a = p[0], b = p[1];
        d[0] = (int)a.x; d[1] = (int)a.y; d[2] = (int)a.z; d[3] = (int)a.w;
        d[4] = (int)b.x; d[5] = (int)b.y; d[6] = (int)b.z; d[7] = (int)b.w;
    } else {
        const uint4 v = *reinterpret_cast<const uint4 *>(digits + at);
        const uint32_t word[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = (int)(int16_t)(word[k >> 1] >> (16 * (k & 1)));
    }
}

// Wide window: which chunk a workgroup takes.  A (chunk, coarse bin) run is only 8 entries there (8192 positions into
// 1024 bins), so a bin's region is written in 32-byte pieces, chunk after chunk.  Workgroups are dealt to the XCDs
// round robin; handing XCD x the chunks [x J/8, (x+1) J/8) in order makes neighbouring pieces come from the SAME
// L2 at about the same time, where they merge into whole lines before they leave for HBM.
__device__ __forceinline__ int sort_xcd_chunk(int x, int J, int xcd) {
    if (!xcd) return x;
    const int per = (J + 7) / 8;
    return (x & 7) * per + (x >> 3);
}

// ---- coarse histogram per (chunk, window) ---------------------------------------------------
// digit d != 0 lands in bucket b = |d| - 1 in [0, nb); coarse bin = b >> LB, fine bucket = b & (2^LB - 1)
template <bool WIDE>
__global__ void __launch_bounds__(SORT_BLOCK)
k_sort_hist1(const int16_t *__restrict__ digits, size_t n_pad, int NC, int LB, int top_row, int period, int LB_top,
             int J, int xcd, uint32_t *__restrict__ hist1, uint32_t *__restrict__ ctrl) {
    extern __shared__ uint32_t lds[];
    const int j = WIDE ? sort_xcd_chunk(blockIdx.x, J, xcd) : (int)blockIdx.x, w = blockIdx.y;
    if (w % period == top_row) LB = LB_top;      // rows of a batch repeat with period = windows per commitment
    // ctrl[0] = #split buckets, [1] = #tasks, [2] = #partial sums, [3] = #big bins, [4] = #buckets split into more
    // than MSM_FINISH_SERIAL segments, [16 ..) = tasks per (length class, window)
    if (blockIdx.x == 0 && w == 0)
        for (int i = threadIdx.x; i < 16 + MSM_SEG * (int)gridDim.y; i += SORT_BLOCK) ctrl[i] = 0;
    if (WIDE && j >= J) return;                  // (the grid is 8 * ceil(J / 8) wide)
    for (int b = threadIdx.x; b < NC; b += SORT_BLOCK) lds[b] = 0;
    __syncthreads();
    const size_t i0 = (size_t)j * SORT_T + 8 * (size_t)threadIdx.x;
    if (i0 < n_pad) {
        int d[8];
        sort_load8<WIDE>(digits, (size_t)w * n_pad + i0, d);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (d[k] != 0) atomicAdd(&lds[((uint32_t)(d[k] < 0 ? -d[k] : d[k]) - 1u) >> LB], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NC; b += SORT_BLOCK) hist1[((size_t)w * NC + b) * J + j] = lds[b];
}

// ---- partition a chunk by coarse bin in LDS; every bin's run leaves contiguous -----------------
template <bool WIDE>
__global__ void __launch_bounds__(SORT_BLOCK)
k_sort_part1(const int16_t *__restrict__ digits, size_t n_pad, int NC, int LB, int top_row, int period, int LB_top,
             int J, int idx_bits, int chunks_per_row, int xcd, const uint32_t *__restrict__ gbase,
             uint32_t *__restrict__ out) {
    extern __shared__ uint32_t lds[];
    uint32_t *cnt = lds;                 // [NC]  run lengths
    uint32_t *lbase = lds + NC;          // [NC]  run starts inside the stage
    uint32_t *gb = lds + 2 * NC;         // [NC]  where the run goes in global memory
    uint32_t *scratch = lds + 3 * NC;    // [16]
    uint32_t *stage = lds + 3 * NC + 16; // [SORT_T]
    // (a persistent form - two workgroups per CU walking the items - measured slower: 52 vs 38 us)
    const int j = WIDE ? sort_xcd_chunk(blockIdx.x, J, xcd) : (int)blockIdx.x, w = blockIdx.y;
    if (WIDE && j >= J) return;
    if (w % period == top_row) LB = LB_top;
    for (int b = threadIdx.x; b < NC; b += SORT_BLOCK) {
        cnt[b] = 0;
        gb[b] = gbase[((size_t)w * NC + b) * J + j];   // strided 4-byte loads: issued first, used last
    }
    __syncthreads();
    const size_t i0 = (size_t)j * SORT_T + 8 * (size_t)threadIdx.x;
    // wide window: the chunk lies inside one table row; its entries carry the COLUMN (k_sort_fine reads the row off the
    // entry's place)
    const size_t e0 = chunks_per_row ? i0 - (size_t)(j / chunks_per_row) * chunks_per_row * SORT_T : i0;
    uint32_t tag[8];     // coarse bin << 16 | rank inside the bin's run; 0xffffffff = no entry
    int d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i0 < n_pad) sort_load8<WIDE>(digits, (size_t)w * n_pad + i0, d);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        tag[k] = 0xffffffffu;
        if (d[k] != 0) {
            const uint32_t cb = ((uint32_t)(d[k] < 0 ? -d[k] : d[k]) - 1u) >> LB;
            tag[k] = (cb << 16) | atomicAdd(&cnt[cb], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the run lengths, 1024 bins per pass
    {
        uint32_t running = 0;
        for (int b0 = 0; b0 < NC; b0 += SORT_BLOCK) {
            const int b = b0 + (int)threadIdx.x;
            uint32_t v = b < NC ? cnt[b] : 0u, tot;
            uint32_t ex = vmpc_block_excl_scan<uint32_t>(v, &tot, scratch);
            if (b < NC) lbase[b] = running + ex;
            running += tot;
        }
    }
    __syncthreads();
    // idx_bits == 31: the fine bucket does not fit beside the index; k_sort_fine re-reads the digit
    const uint32_t fmask = idx_bits < 31 ? (1u << LB) - 1u : 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (tag[k] != 0xffffffffu) {
            const uint32_t b = (uint32_t)(d[k] < 0 ? -d[k] : d[k]) - 1u;
            const uint32_t cb = tag[k] >> 16, pos = lbase[cb] + (tag[k] & 0xffffu);
            stage[pos] = (uint32_t)(e0 + k) | ((b & fmask) << (idx_bits & 31)) | (d[k] < 0 ? 0x80000000u : 0u);
            // wide window: a bin's run of the chunk is ~8 entries; a wave walking 64 such runs one after the other
            // (below) keeps 8 of its lanes busy for 64 LDS round trips - every staged entry gets its destination
            // instead and the stage leaves in one sweep
            if (WIDE) stage[SORT_T + pos] = gb[cb] + (tag[k] & 0xffffu);
        }
    }
    __syncthreads();
    if (WIDE) {
        const uint32_t total = lbase[NC - 1] + cnt[NC - 1];
        for (uint32_t s = threadIdx.x; s < total; s += SORT_BLOCK) out[stage[SORT_T + s]] = stage[s];
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < NC; b += SORT_WAVES) {
        const uint32_t len = cnt[b], lb = lbase[b], g = gb[b];
        for (uint32_t k = lane; k < len; k += 64) out[g + k] = stage[lb + k];
    }
}

// ---- fine sort of one (coarse bin, window): counts, starts, bucket-sorted indices -----------------
template <bool FINE_IN_ENTRY>
__device__ __forceinline__ uint32_t sort_fine_of(uint32_t e, int idx_bits, uint32_t fmask,
                                                 const int16_t *__restrict__ drow) {
    if (FINE_IN_ENTRY) return (e >> idx_bits) & fmask;
    const int d = (int)drow[e & 0x7fffffffu];       // huge index spaces (fixed-base tables beyond 2^22 entries
    return ((uint32_t)(d < 0 ? -d : d) - 1u) & fmask;   // per fine bit): one 2-byte gather per entry instead
}

// wide window: the table row an entry belongs to, read off its place `at` in the coarse bin's region (the runs of a
// bin are laid out chunk by chunk and a chunk lies inside one row: rb[r] = where row r's runs begin), times the row
// stride: what turns the entry's column into the table position.  n_rows == 0 (every other plan): nothing to add.
__device__ __forceinline__ uint32_t sort_row_offset(const uint32_t *rb, int n_rows, uint32_t at, uint32_t row_stride) {
    uint32_t row = 0;
    for (int r = 1; r < n_rows; r++) row += rb[r] <= at ? 1u : 0u;
    return row * row_stride;
}

// LB of a row, and (top row only) the buckets no digit of a canonical scalar reaches are marked empty
__device__ __forceinline__ int sort_row_lb(int w, int cb, int NC, int LB, int top_row, int period, int LB_top,
                                           int nb1, uint32_t *__restrict__ counts, uint32_t *__restrict__ nseg) {
    if (w % period != top_row) return LB;
    if (counts) {
        const uint32_t covered = (uint32_t)NC << LB_top, nb = (uint32_t)nb1 - 1u;
        for (uint32_t u = covered + cb * SORT_BLOCK + threadIdx.x; u < nb; u += NC * SORT_BLOCK) {
            counts[(size_t)w * nb1 + 1 + u] = 0;
            nseg[(size_t)w * nb1 + 1 + u] = 0;
        }
    }
    return LB_top;
}

// One workgroup per (coarse bin, window).  A bin of up to SORT_FINE_CAP entries (every bin of uniformly
// distributed scalars) lives in registers between the two phases: one read of the entries, one LDS atomic
// each - whose return value IS the entry's rank inside its bucket - then the bucket-sorted order is
// assembled in LDS and leaves as whole lines (4-byte stores scattered over the bin's 32-KiB region cost one
// L2 request each: 135 us at n = 2^20; LDS takes them at bank rate).  Larger bins (skewed witnesses, the
// under-full top window of some plans) are only counted here and sorted by k_sort_fine_big.
// The tail is pass 1 of the segment planning: segments per bucket, the block's histogram of segment lengths,
// and partial-sum slots for buckets that need several segments.
template <bool FINE_IN_ENTRY, int CAP>
__global__ void __launch_bounds__(SORT_BLOCK, 8)      // 8 waves per SIMD = two workgroups per CU (<= 64 VGPRs)
k_sort_fine(const uint32_t *__restrict__ in, const uint32_t *__restrict__ gbase, int NC, int W, int LB, int top_row,
            int period, int LB_top, int J, int idx_bits, int chunks_per_row, uint32_t row_stride, int nb1,
            const int16_t *__restrict__ digits, size_t n_pad,
            uint32_t *__restrict__ counts, uint32_t *__restrict__ starts, uint32_t *__restrict__ sorted,
            int seg_shift, int balanced, uint32_t *__restrict__ nseg, uint32_t *__restrict__ block_hist,
            uint32_t *__restrict__ heavy_list, uint32_t *__restrict__ seg_starts,
            uint32_t *__restrict__ ctrl /*[0] = #split buckets, [2] = #partial sums, [3] = #big bins*/) {
    __shared__ uint32_t cnt[512], cur[512], scratch[16];
    __shared__ uint32_t lh[MSM_SEG + 1], heavy_n, heavy_segs, heavy_base, heavy_seg_base;
    __shared__ uint32_t rb[MSM_WIDE_ROWS + 3];         // wide window: where each table row's entries start in this bin
    constexpr int REG = CAP / SORT_BLOCK;              // entries per thread of a staged bin
    uint32_t *stage;
    if constexpr (CAP > SORT_FINE_CAP) {               // (the wide window's 64 KB stage: dynamic LDS)
        extern __shared__ uint32_t fine_stage_dyn[];
        stage = fine_stage_dyn;
    } else {
        __shared__ uint32_t fine_stage[SORT_FINE_CAP];
        stage = fine_stage;
    }
    // top window first: under-full, so its bins are the fullest
    const int cb = blockIdx.x % NC, w = W - 1 - blockIdx.x / NC;
    LB = sort_row_lb(w, cb, NC, LB, top_row, period, LB_top, nb1, counts, nseg);
    const int NF = 1 << LB;
    const size_t slot = (size_t)w * NC + cb;
    const uint32_t lo = gbase[slot * J], hi = gbase[(slot + 1) * J];   // gbase[H] = total
    const uint32_t fmask = (uint32_t)NF - 1u, imask = FINE_IN_ENTRY ? (1u << idx_bits) - 1u : 0x7fffffffu;
    const int16_t *drow = digits + (size_t)w * n_pad;
    const int lane = threadIdx.x & 63;
    const bool staged = hi - lo <= (uint32_t)CAP;
    const int n_rows = chunks_per_row ? J / chunks_per_row : 0;
    for (int f = threadIdx.x; f < NF; f += SORT_BLOCK) cnt[f] = 0;
    if (threadIdx.x <= MSM_SEG) lh[threadIdx.x] = 0;
    if (threadIdx.x == 0) heavy_n = heavy_segs = 0;
    if ((int)threadIdx.x < n_rows) rb[threadIdx.x] = gbase[slot * J + (size_t)threadIdx.x * chunks_per_row];
    __syncthreads();
    uint32_t ev[REG], rk2[REG / 2];      // ranks < 2^16, two per register
    if (staged) {
#pragma unroll
        for (int k = 0; k < REG; k++) {
            const uint32_t i = lo + k * SORT_BLOCK + threadIdx.x;
            ev[k] = i < hi ? in[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < REG; k++) {
            uint32_t r = 0;
            if (lo + k * SORT_BLOCK + threadIdx.x < hi)
                r = atomicAdd(&cnt[sort_fine_of<FINE_IN_ENTRY>(ev[k], idx_bits, fmask, drow)], 1u);
            rk2[k >> 1] = (k & 1) ? (rk2[k >> 1] | (r << 16)) : r;
        }
    } else {
        for (uint32_t i = lo + threadIdx.x; i < hi; i += SORT_BLOCK) {
            const uint32_t f = sort_fine_of<FINE_IN_ENTRY>(in[i], idx_bits, fmask, drow);
            // skewed witnesses (mostly 0 / 1) put a whole wave's entries into one bucket: one atomic per wave
            const unsigned long long act = __ballot(1);
            const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
            if (__ballot(f == f0) == act) {
                if (lane == (int)__builtin_ctzll(act)) atomicAdd(&cnt[f0], (uint32_t)__popcll(act));
            } else {
                atomicAdd(&cnt[f], 1u);
            }
        }
    }
    __syncthreads();
    {
        uint32_t v = threadIdx.x < (unsigned)NF ? cnt[threadIdx.x] : 0u, tot;
        uint32_t ex = vmpc_block_excl_scan<uint32_t>(v, &tot, scratch);
        const size_t ci = (size_t)w * nb1 + 1 + (size_t)cb * NF + threadIdx.x;   // bucket |d| = b + 1
        if (threadIdx.x < (unsigned)NF) {
            counts[ci] = v;
            starts[ci] = lo + ex;
            cur[threadIdx.x] = ex;
        }
        if (cb == 0 && threadIdx.x == 0) {       // slot 0 of a window is nobody's bucket
            counts[(size_t)w * nb1] = 0;
            starts[(size_t)w * nb1] = lo;
            nseg[(size_t)w * nb1] = 0;
        }
        const uint32_t seg_log = (uint32_t)(MSM_SEG_LOG2 + seg_shift);
        const uint32_t ns = msm_seg_count(v, seg_log);            // equal segments (msm_seg_range)
        uint32_t my_heavy = 0, my_seg = 0;
        if (threadIdx.x < (unsigned)NF) {
            nseg[ci] = ns;
            if (balanced) {
                if (ns) atomicAdd(&lh[msm_seg_class((v + ns - 1u) / ns, seg_shift)], ns);
            } else {
                const uint32_t full = v >> seg_log, rem = v & ((1u << seg_log) - 1u);
                if (full) atomicAdd(&lh[MSM_SEG], full);
                if (rem) atomicAdd(&lh[msm_seg_class(rem, seg_shift)], 1u);
            }
            if (ns > 1) {
                my_heavy = atomicAdd(&heavy_n, 1u);
                my_seg = atomicAdd(&heavy_segs, ns);
                if (ns > MSM_FINISH_SERIAL) atomicAdd(&ctrl[4], 1u);     // rare: skewed scalars only
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && heavy_n) {       // slots handed out by one global atomic per workgroup
            heavy_base = atomicAdd(&ctrl[0], heavy_n);
            heavy_seg_base = atomicAdd(&ctrl[2], heavy_segs);
        }
        if (threadIdx.x == 0 && !staged) atomicAdd(&ctrl[3], 1u);
        __syncthreads();
        if (threadIdx.x < (unsigned)NF && ns > 1) {
            heavy_list[heavy_base + my_heavy] = (uint32_t)ci;
            seg_starts[ci] = heavy_seg_base + my_seg;
        }
        // this block's segments per length class, one 256-byte line; k_msm_ranks turns the column of a
        // window's blocks into first ranks
        if (threadIdx.x >= 1 && threadIdx.x <= MSM_SEG)
            block_hist[slot * MSM_SEG + (threadIdx.x - 1)] = lh[threadIdx.x];
    }
    if (!staged) return;
#pragma unroll
    for (int k = 0; k < REG; k++) {
        const uint32_t at = lo + k * SORT_BLOCK + threadIdx.x;
        if (at < hi) {
            const uint32_t e = ev[k];
            const uint32_t f = sort_fine_of<FINE_IN_ENTRY>(e, idx_bits, fmask, drow);
            stage[cur[f] + ((rk2[k >> 1] >> (16 * (k & 1))) & 0xffffu)] =
                ((e & imask) + sort_row_offset(rb, n_rows, at, row_stride)) | (e & 0x80000000u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < hi - lo; i += SORT_BLOCK) sorted[lo + i] = stage[i];
}

// Bins beyond the stage: tiles of SORT_FINE_CAP entries, each sorted in LDS by fine bucket with tile-local
// ranks; a bucket's run of the tile then leaves as one contiguous piece (one wave per bucket; the whole
// workgroup for a run that dominates the tile), appended at the bucket's cursor.  Launched over all bins;
// a workgroup whose bin was staged (or when no bin was big: ctrl[3] == 0) exits at once.
template <bool FINE_IN_ENTRY, int CAP>
__global__ void __launch_bounds__(SORT_BLOCK)
k_sort_fine_big(const uint32_t *__restrict__ in, const uint32_t *__restrict__ gbase, int NC, int W, int LB,
                int top_row, int period, int LB_top, int J, int idx_bits, int chunks_per_row, uint32_t row_stride, int nb1,
                const int16_t *__restrict__ digits, size_t n_pad, const uint32_t *__restrict__ starts,
                uint32_t *__restrict__ sorted, const uint32_t *__restrict__ ctrl) {
    // (a small grid walking the slots instead of one workgroup per slot was tried in round 4 to make the empty case
    // cheaper: 15 against 13 us - the cost of this launch is not its workgroup count)
    if (ctrl[3] == 0) return;
    const int cb = blockIdx.x % NC, w = W - 1 - blockIdx.x / NC;
    const size_t slot = (size_t)w * NC + cb;
    const uint32_t lo = gbase[slot * J], hi = gbase[(slot + 1) * J];
    if (hi - lo <= (uint32_t)CAP) return;
    __shared__ uint32_t tcnt[512], tex[512], cur[512], scratch[16], long_runs[CAP / 512 + 1], n_long;
    __shared__ uint32_t rb[MSM_WIDE_ROWS + 3];
    constexpr int REG = CAP / SORT_BLOCK;
    uint32_t *stage;
    if constexpr (CAP > SORT_FINE_CAP) {
        extern __shared__ uint32_t fine_stage_dyn[];
        stage = fine_stage_dyn;
    } else {
        __shared__ uint32_t fine_stage[SORT_FINE_CAP];
        stage = fine_stage;
    }
    const int n_rows = chunks_per_row ? J / chunks_per_row : 0;
    if ((int)threadIdx.x < n_rows) rb[threadIdx.x] = gbase[slot * J + (size_t)threadIdx.x * chunks_per_row];
    LB = sort_row_lb(w, cb, NC, LB, top_row, period, LB_top, nb1, nullptr, nullptr);
    const int NF = 1 << LB;
    const uint32_t fmask = (uint32_t)NF - 1u, imask = FINE_IN_ENTRY ? (1u << idx_bits) - 1u : 0x7fffffffu;
    const int16_t *drow = digits + (size_t)w * n_pad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < (unsigned)NF) cur[threadIdx.x] = starts[(size_t)w * nb1 + 1 + (size_t)cb * NF + threadIdx.x];
    uint32_t ev[REG], rk2[REG / 2];
    for (uint32_t t0 = lo; t0 < hi; t0 += CAP) {
        const uint32_t t1 = t0 + CAP < hi ? t0 + CAP : hi;
        for (int f = threadIdx.x; f < NF; f += SORT_BLOCK) tcnt[f] = 0;
        if (threadIdx.x == 0) n_long = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < REG; k++) {
            const uint32_t i = t0 + k * SORT_BLOCK + threadIdx.x;
            ev[k] = i < t1 ? in[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < REG; k++) {
            uint32_t r = 0;
            if (t0 + k * SORT_BLOCK + threadIdx.x < t1) {
                const uint32_t f = sort_fine_of<FINE_IN_ENTRY>(ev[k], idx_bits, fmask, drow);
                const unsigned long long act = __ballot(1);
                const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
                if (__ballot(f == f0) == act) {          // a wave of one bucket (skew): one atomic
                    uint32_t base = 0;
                    if (lane == (int)__builtin_ctzll(act)) base = atomicAdd(&tcnt[f0], (uint32_t)__popcll(act));
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    r = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
                } else {
                    r = atomicAdd(&tcnt[f], 1u);
                }
            }
            rk2[k >> 1] = (k & 1) ? (rk2[k >> 1] | (r << 16)) : r;
        }
        __syncthreads();
        {
            uint32_t v = threadIdx.x < (unsigned)NF ? tcnt[threadIdx.x] : 0u, tot;
            uint32_t ex = vmpc_block_excl_scan<uint32_t>(v, &tot, scratch);
            if (threadIdx.x < (unsigned)NF) tex[threadIdx.x] = ex;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < REG; k++) {
            const uint32_t at = t0 + k * SORT_BLOCK + threadIdx.x;
            if (at < t1) {
                const uint32_t e = ev[k];
                const uint32_t f = sort_fine_of<FINE_IN_ENTRY>(e, idx_bits, fmask, drow);
                stage[tex[f] + ((rk2[k >> 1] >> (16 * (k & 1))) & 0xffffu)] =
                    ((e & imask) + sort_row_offset(rb, n_rows, at, row_stride)) | (e & 0x80000000u);
            }
        }
        __syncthreads();
        for (int f = wave; f < NF; f += SORT_WAVES) {
            const uint32_t len = tcnt[f], src = tex[f], dst = cur[f];
            if (len > 512) {                            // a bucket that dominates the tile: whole workgroup, below
                if (lane == 0) long_runs[atomicAdd(&n_long, 1u)] = (uint32_t)f;
                continue;
            }
            for (uint32_t k = lane; k < len; k += 64) sorted[dst + k] = stage[src + k];
            if (lane == 0) cur[f] = dst + len;
        }
        __syncthreads();
        for (uint32_t r = 0; r < n_long; r++) {
            const uint32_t f = long_runs[r];
            const uint32_t len = tcnt[f], src = tex[f], dst = cur[f];
            for (uint32_t k = threadIdx.x; k < len; k += SORT_BLOCK) sorted[dst + k] = stage[src + k];
        }
        __syncthreads();
        if (threadIdx.x < n_long) cur[long_runs[threadIdx.x]] += tcnt[long_runs[threadIdx.x]];
        __syncthreads();
    }
}

// block_hist[block][class] (segments of that length in the block) -> the block's first rank among its
// window's segments of that class, in place; the window's class totals go to ctrl[16 + class * W + w].
// One workgroup per window: thread (class, group) walks NC / 16 consecutive blocks.
__device__ __forceinline__ void msm_ranks_window(uint32_t *__restrict__ block_hist, int NC, int W, int w,
                                                 uint32_t *__restrict__ ctrl, uint32_t (*part)[MSM_SEG]) {
    const int cls = threadIdx.x % MSM_SEG, g = threadIdx.x / MSM_SEG;      // 16 groups
    const int per = (NC + 15) / 16, b0 = g * per, b1 = b0 + per < NC ? b0 + per : NC;
    uint32_t run = 0;
    for (int b = b0; b < b1; b++) run += block_hist[((size_t)w * NC + b) * MSM_SEG + cls];
    part[g][cls] = run;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int k = 0; k < 16; k++) {
        const uint32_t v = part[k][cls];
        if (k < g) before += v;
        total += v;
    }
    run = before;
    for (int b = b0; b < b1; b++) {
        const size_t at = ((size_t)w * NC + b) * MSM_SEG + cls;
        const uint32_t v = block_hist[at];
        block_hist[at] = run;
        run += v;
    }
    // class index cls = L - 1; cells are ordered longest class first
    if (g == 0) ctrl[16 + (MSM_SEG - 1 - cls) * W + w] = total;
}

__global__ void __launch_bounds__(1024)
k_msm_ranks(uint32_t *__restrict__ block_hist, int NC, int W, uint32_t *__restrict__ ctrl) {
    __shared__ uint32_t part[16][MSM_SEG];
    msm_ranks_window(block_hist, NC, W, (int)blockIdx.x, ctrl, part);
}

// first task id of every (length class, window) cell: exclusive scan of the 64 x W totals in the order
// longest class first, windows ascending - longest segments get the smallest task ids
__device__ __forceinline__ void msm_classes_body(uint32_t *__restrict__ ctrl, int W, uint32_t *__restrict__ class_base,
                                                 uint32_t *scratch) {
    const int ncell = MSM_SEG * W;                  // <= 4096
    uint32_t running = 0;
    for (int k0 = 0; k0 < ncell; k0 += 1024) {
        const int k = k0 + (int)threadIdx.x;
        uint32_t v = k < ncell ? ctrl[16 + k] : 0u, tot;
        uint32_t ex = vmpc_block_excl_scan<uint32_t>(v, &tot, scratch);
        if (k < ncell) class_base[k] = running + ex;
        running += tot;
    }
    if (threadIdx.x == 0) ctrl[1] = running;        // #tasks
}

__global__ void __launch_bounds__(1024)
k_msm_classes(uint32_t *__restrict__ ctrl, int W, uint32_t *__restrict__ class_base) {
    __shared__ uint32_t scratch[16];
    msm_classes_body(ctrl, W, class_base, scratch);
}

// few windows (one commitment over a tabulated vector, a prover round's pair): both steps in one workgroup, the
// windows one after the other - a launch less
__global__ void __launch_bounds__(1024)
k_msm_ranks_classes(uint32_t *__restrict__ block_hist, int NC, int W, uint32_t *__restrict__ ctrl,
                    uint32_t *__restrict__ class_base) {
    __shared__ uint32_t part[16][MSM_SEG];
    __shared__ uint32_t scratch[16];
    for (int w = 0; w < W; w++) {
        msm_ranks_window(block_hist, NC, W, w, ctrl, part);
        __syncthreads();
    }
    msm_classes_body(ctrl, W, class_base, scratch);
}

// plan, pass 2 (pass 1 is the tail of k_sort_fine): one block per (window, coarse bin), one thread per
// bucket: task id = first id of the (length, window) cell + this block's first rank in it + a local rank
__global__ void __launch_bounds__(512)
k_msm_plan2(const uint32_t *__restrict__ counts, int NC, int W, int NF, int top_row, int period, int NF_top, int nb1,
            int seg_shift, int balanced,
            const uint32_t *__restrict__ class_base, const uint32_t *__restrict__ block_rank,
            uint2 *__restrict__ tasks) {
    const uint32_t seg_log = (uint32_t)(MSM_SEG_LOG2 + seg_shift);
    __shared__ uint32_t cur[MSM_SEG + 1], first[MSM_SEG + 1];
    const uint32_t block = blockIdx.x;
    const int w = block / NC, cb = block % NC;
    if (threadIdx.x >= 1 && threadIdx.x <= MSM_SEG) {
        cur[threadIdx.x] = 0;
        first[threadIdx.x] = class_base[(MSM_SEG - threadIdx.x) * W + w] +
                             block_rank[(size_t)block * MSM_SEG + (threadIdx.x - 1)];
    }
    __syncthreads();
    if (w % period == top_row) NF = NF_top;
    if (threadIdx.x >= (unsigned)NF) return;
    const size_t ci = (size_t)w * nb1 + 1 + (size_t)cb * NF + threadIdx.x;
    const uint32_t cnt = counts[ci];
    const uint32_t ns = msm_seg_count(cnt, seg_log);
    if (balanced) {
        if (ns) {                               // the bucket's ns equal segments are consecutive tasks of one class
            const uint32_t bin = msm_seg_class((cnt + ns - 1u) / ns, seg_shift);
            const uint32_t base = first[bin] + atomicAdd(&cur[bin], ns);
            for (uint32_t sidx = 0; sidx < ns; sidx++) tasks[base + sidx] = make_uint2((uint32_t)ci, sidx);
        }
        return;
    }
    const uint32_t full = cnt >> seg_log, rem = cnt & ((1u << seg_log) - 1u);
    if (rem) {
        const uint32_t bin = msm_seg_class(rem, seg_shift);
        tasks[first[bin] + atomicAdd(&cur[bin], 1u)] = make_uint2((uint32_t)ci, full);
    }
    if (full) {
        const uint32_t base = first[MSM_SEG] + atomicAdd(&cur[MSM_SEG], full);
        for (uint32_t sidx = 0; sidx < full; sidx++) tasks[base + sidx] = make_uint2((uint32_t)ci, sidx);
    }
}

// ---- host side ---------------------------------------------------------------------------
// Window width.  Measured on MI355X (scripts/window_sweep.py), not modelled: the tail stages
// (reduce, recombination) are latency chains whose length barely depends on c, so the widest
// window the int16 digits allow wins as soon as the bucket stage matters (n > 2^13); below that
// c = 11 keeps the reduce short.  Both choices also leave the top window of a 253-bit scalar
// empty or well spread (W*c = 264 resp. 256), where other widths pile n/2 entries into one bucket.
static int msm_pick_window(size_t n, int scalar_bits) {
    if (scalar_bits == 253) return n > (1u << 13) ? 16 : 11;   // Ed25519; BN-256 keeps the model
    double best = 1e300;
    int best_c = 4;
    for (int c = 4; c <= MSM_MAX_C; c++) {
        int W = (scalar_bits + 2 + c - 1) / c;
        double cost = (double)W * ((double)n + 2.5 * (double)(1u << (c - 1)));
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    return best_c;
}

// largest bucket index (|digit| - 1, after the signed recoding's carry) the top window can hold for
// canonical scalars: (modulus - 1) >> (c * (W - 1)), plus the carry, minus one
static uint32_t msm_top_max_bucket(const msm_modulus &mod, int c, int W) {
    uint32_t m1[8];
    uint64_t borrow = 1;
    for (int i = 0; i < 8; i++) {
        uint64_t v = (uint64_t)mod.v[i] - borrow;
        m1[i] = (uint32_t)v;
        borrow = (v >> 63) & 1;
    }
    const int shift = c * (W - 1);
    if (shift >= 256) return 0;
    uint64_t top = 0;
    for (int bit = 0; bit < 32 && shift + bit < 256; bit++)
        top |= (uint64_t)((m1[(shift + bit) >> 5] >> ((shift + bit) & 31)) & 1u) << bit;
    for (int bit = shift + 32; bit < 256; bit++)
        if ((m1[bit >> 5] >> (bit & 31)) & 1u) return 0xffffffffu;
    return (uint32_t)(top > 0xfffffffeull ? 0xffffffffull : top);     // raw + carry - 1 = raw
}

void msm_make_plan(vmpc_ctx *ctx, size_t n_main, size_t n_extra, int scalar_bits, msm_plan &p,
                   const msm_modulus *modulus) {
    p.n_main = n_main;
    p.n_extra = n_extra;
    p.n_total = n_main + n_extra;
    p.scalar_bits = scalar_bits;
    p.c = ctx->window_override ? ctx->window_override : msm_pick_window(p.n_total, scalar_bits);
    if (p.c < 4) p.c = 4;
    if (p.c > MSM_MAX_C) p.c = MSM_MAX_C;
    // the signed recoding may carry one bit past the top: W * c >= scalar_bits + 2 keeps the top
    // window's raw digit below 2^(c-1) (253-bit Ed25519 scalars: W = ceil(255 / c))
    p.W = (scalar_bits + 2 + p.c - 1) / p.c;
    while (p.W > 64) {   // the recombination kernels give one lane to each window (64-lane wave)
        p.c++;
        p.W = (scalar_bits + 2 + p.c - 1) / p.c;
    }
    p.top_row = -1;
    p.top_max_b = 0;
    p.period = p.W;
    if (modulus) {
        p.top_row = p.W - 1;
        p.top_max_b = msm_top_max_bucket(*modulus, p.c, p.W);
    }
    msm_plan_geometry(ctx, p);
}

// everything that follows from (n_total, c, W): sort decomposition, segment length, reduce shape
void msm_plan_geometry(vmpc_ctx *ctx, msm_plan &p) {
    if (p.period <= 0 || p.period > p.W) p.period = p.W;
    p.nb = 1 << (p.c - 1);
    p.nb1 = p.nb + 1;
    p.n_pad = (p.n_total + 7) & ~(size_t)7;
    // two-level sort: bucket b = coarse << LB | fine.  The fine buckets of a coarse bin are resolved by
    // one workgroup in LDS (<= 512), the index and the fine bucket share a 32-bit entry with the sign,
    // and a coarse bin should hold ~8 K entries (its workgroup reads it twice out of L2) - but not be
    // so narrow that a chunk's run of a bin (8192 / NC entries) drops below a quarter line.
    p.idx_bits = 1;
    while (((size_t)1 << p.idx_bits) < p.n_pad) p.idx_bits++;
    // (Rows of 2^22 positions and more - the digit rows of a 4- / 8- / 16-row table over a large CRS - stop at 256
    // coarse bins: the histogram and partition kernels scan and scatter per BIN, the fine sort's tiled path takes the
    // 16-32 K entries of such a bin in its stride; round 5, scripts/pair_sort_probe.py: one commitment over the 4-row
    // table at 2^20 1.187 -> 1.17 ms, a prover round's pair over the 8-row table 1.18 -> 1.16 ms, 2^21 over 4 rows
    // 2.09 -> 2.02 ms; 128 bins are better still for the pair and worse for a single vector, 64 are 0.4 ms worse.)
    const int nc_target = p.n_total >= ((size_t)1 << 22) ? 256 : 512;
    int lb = p.c - 1;
    if (lb > 9) lb = 9;
    while (lb > 0 && (p.nb >> lb) < nc_target && (p.n_total >> (p.c - 1 - lb)) > 8192) lb--;
    if (ctx->sort_fine_bits >= 0 && ctx->sort_fine_bits <= 9 && ctx->sort_fine_bits <= p.c - 1 &&
        (p.nb >> ctx->sort_fine_bits) <= 4096)
        lb = ctx->sort_fine_bits;                                                          // tuning knob
    p.fine_cap = SORT_FINE_CAP;
    if (p.wide) {
        // one set of 2^19 buckets fed by 13 table rows: 1024 coarse bins of 512 buckets; at 2^20 columns a bin holds
        // ~13.3 K entries (sigma 115), staged whole in a 16 K-entry LDS stage.  The entry between the passes carries the
        // column and the fine bucket; the table row comes from the entry's place (k_sort_fine, sort_row_offset)
        // (beyond 2^20 columns - a 2^21-generator CRS, one GPU's share of BASELINE config 4 - narrower bins keep them
        // inside the stage: 2048 bins of 256 buckets)
        lb = 9;
        while (lb > 6 && (p.n_total >> (p.c - 1 - lb)) > 14336) lb--;
        p.idx_bits = p.col_bits;
        p.fine_cap = 16384;
    }
    p.LB = lb;
    p.NC = p.nb >> lb;                                       // <= 4096 (c <= 16, lb >= 3 whenever nb > 4096)
    p.fine_in_entry = p.idx_bits + lb <= 31;
    // the top window alone in its row (plain MSMs, one-row tables): its digits stop at top_max_b, so its
    // coarse bins are cut finer - the smallest LB_top that still maps every reachable bucket below NC
    p.LB_top = lb;
    if (p.top_row >= 0) {
        int lt = 0;
        while (lt < lb && (p.top_max_b >> lt) > (uint32_t)(p.NC - 1)) lt++;
        p.LB_top = lt;
    }
    p.J = (int)((p.n_pad + SORT_T - 1) / SORT_T);
    // segment length: the bucket stage wants ~4 tasks per lane of the chip (2^18) of equal length.
    // 64 entries up to W * n = 2^24 (n = 2^20 at c = 16), doubled from there - otherwise every
    // bucket of a 2^22-term MSM is split in three and the finish stage (one more gather of
    // 160-byte partial sums) costs 14 %
    // (work = the entries to expect: every position of every digit row, unless the caller knows the rows to be
    // sparsely populated - with the positions of a half-empty pair the prover's rounds got 128-entry segments,
    // 2^17 tasks for 2^18 lanes, and a bucket stage at half occupancy: 0.97 instead of 0.68 ms)
    p.balanced = p.scalar_bits == 253 ? 1 : 0;       // Ed25519 (4 waves per SIMD) / BN-256 (1 wave per SIMD): msm_sort.h
    const size_t work = ((size_t)p.W * p.n_total) >> ctx->plan_fill_shift;
    p.seg_shift = 0;
    while (p.seg_shift < 4 && (((size_t)MSM_SEG << p.seg_shift) << 18) < work) p.seg_shift++;
    // ... and SHORTER ones for short inputs: with fewer tasks than lanes the stage lasts as long as its longest
    // chain, i.e. the fullest bucket (Poisson tail: 22 entries at a mean of 8 = 93 us in the prover's late
    // rounds).  Halving the segments keeps >= 2^18 tasks down to 8-entry segments; the partial sums of split
    // buckets go through the finish kernels.
    while (p.seg_shift > ctx->seg_shift_min && (((size_t)MSM_SEG >> -(p.seg_shift - 1)) << 18) >= work)
        p.seg_shift--;
    // reduce: chunk-lanes per window (chunk length a power of two): the per-lane work is a
    // dependency chain, so shorter chunks on more lanes cut the latency
    // (with few windows - fixed-base tables - more chunk-lanes per window keep the same ~64 K lanes busy)
    int chunks = MSM_REDUCE_CHUNKS;
    // (one wide-window commitment: 2^16 chunk-lanes of 8 buckets, 256 workgroups - the whole chip - instead of 128)
    while (chunks * 2 * p.W <= MSM_REDUCE_CHUNKS * 16 &&
           chunks * 2 <= (p.wide && ctx->reduce_max_chunks == 32768 ? 65536 : ctx->reduce_max_chunks))
        chunks *= 2;
    // ... and FEWER for many windows (several commitments in one pass): the reduction kernel holds 256 VGPRs, so
    // the chip keeps 2^17 of its lanes resident; beyond 2^16 chunk-lanes a pass runs in rounds, and every lane
    // repeats the offset ladder - at three commitments per pass 1024 chunk-lanes per window do half the work of
    // 4096 (0.93 -> 0.885 ms per commitment with three passes in flight, profiles/r04_probes/reduce_chunks.txt)
    while (chunks * p.W > MSM_REDUCE_CHUNKS * 16 && chunks > 256) chunks /= 2;
    if (ctx->reduce_chunks_override >= 64) chunks = ctx->reduce_chunks_override;        // (experiment / throughput mode)
    if (chunks > p.nb) chunks = p.nb;
    p.chunks = chunks;
    p.chunk_len = p.nb / chunks;
    p.red_blocks = (chunks + MSM_BLOCK - 1) / MSM_BLOCK;
}

void msm_layout(const msm_plan &p, msm_ws &w, char *base, size_t entry_bytes, size_t acc_bytes) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += vmpc_align(bytes);
        return base ? (void *)(base + o) : (void *)nullptr;
    };
    size_t nbk = (size_t)p.W * p.nb1;
    w.entries = (uint32_t *)take(p.n_total * entry_bytes);
    w.digits = (int16_t *)take((size_t)p.W * p.n_pad * (p.wide ? 4 : 2));
    w.hist1_n = (size_t)p.W * p.NC * p.J;
    w.hist1 = (uint32_t *)take((w.hist1_n + 1) * 4);
    w.counts = (uint32_t *)take(nbk * 4);
    w.starts = (uint32_t *)take(nbk * 4);
    w.stage1 = (uint32_t *)take((size_t)p.W * p.n_total * 4);
    w.sorted = (uint32_t *)take((size_t)p.W * p.n_total * 4);
    w.buckets = (uint32_t *)take((size_t)p.W * p.nb * acc_bytes);
    w.partials = (uint32_t *)take((size_t)p.W * (3 * p.red_blocks + 4) * acc_bytes);     // msm_sort.h
    // segment planning: at most M/SEG full segments plus one remainder per non-empty bucket
    size_t m_max = (size_t)p.W * p.n_total;
    size_t nonempty_max = m_max < (size_t)p.W * p.nb ? m_max : (size_t)p.W * p.nb;
    w.t_max = m_max / msm_seg_len(p) + nonempty_max;
    w.plan_blocks = (uint32_t)(p.W * p.NC);                  // one block of the task table per (window, coarse bin)
    size_t hist_n = (size_t)MSM_SEG * w.plan_blocks;
    w.block_hist = (uint32_t *)take(hist_n * 4);              // [block][length class]: the block's first rank in the cell
    w.block_base = (uint32_t *)take((size_t)MSM_SEG * p.W * 4);   // [length class][window]: the cell's first task id
    w.nseg = (uint32_t *)take(nbk * 4);
    w.seg_starts = (uint32_t *)take(nbk * 4);                 // written for split buckets only
    w.heavy_list = (uint32_t *)take(nbk * 4);
    w.ctrl = (uint32_t *)take((16 + (size_t)MSM_SEG * p.W) * 4);
    w.tasks = (uint2 *)take(w.t_max * 8);
    w.seg_partial = (uint32_t *)take(w.t_max * acc_bytes);
    size_t scan_n = hist_n;
    if (w.hist1_n > scan_n) scan_n = w.hist1_n;
    w.scan_ws = take(vmpc_scan_ws_bytes(scan_n, 4));
    w.total = off;
}

int msm_sort_stage(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w, const void *scalars, size_t n,
                   const void *extra_scalars, const msm_modulus &modulus) {
    hipStream_t st = ctx->stream;
    const size_t n_total = p.n_total;
    const unsigned gb = (unsigned)((p.n_pad + MSM_BLOCK - 1) / MSM_BLOCK);
    {
        vmpc_stage_scope s(ctx, "msm_recode");
        k_msm_recode<<<gb, MSM_BLOCK, 0, st>>>((const uint32_t *)scalars, n,
                                              (const uint32_t *)extra_scalars, n, n_total - n, p.n_pad, w.digits, p.c,
                                              p.W, p.W, p.n_pad, modulus, ctx->d_status);
        VMPC_KERNEL_CHECK();
    }
    return msm_sort_digits(ctx, p, w);
}

int msm_recode_rows(vmpc_ctx *ctx, const void *scalars, size_t n_main, const void *extra_scalars,
                    size_t extra_pos, size_t n_extra, size_t n_pad, int16_t *digits, int c, int W, int rows,
                    const msm_modulus &modulus) {
    vmpc_stage_scope s(ctx, "msm_recode");
    const int wpr = W / rows;
    k_msm_recode<<<(unsigned)((n_pad + MSM_BLOCK - 1) / MSM_BLOCK), MSM_BLOCK, 0, ctx->stream>>>(
        (const uint32_t *)scalars, n_main, (const uint32_t *)extra_scalars, extra_pos, n_extra, n_pad, digits, c,
        W, wpr, (size_t)rows * n_pad, modulus, ctx->d_status);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

int msm_recode_rows_batch(vmpc_ctx *ctx, const void *const *scalars, size_t n_main, const void *const *extra_scalars,
                          int K, size_t extra_pos, size_t n_extra, size_t n_pad, int16_t *digits,
                          size_t digits_per_commitment, int c, int W, int rows, const msm_modulus &modulus) {
    vmpc_stage_scope s(ctx, "msm_recode");
    msm_recode_batch b;
    memset(&b, 0, sizeof b);
    for (int k = 0; k < K; k++) {
        b.sc[k] = (const uint32_t *)scalars[k];
        b.sc_extra[k] = extra_scalars ? (const uint32_t *)extra_scalars[k] : nullptr;
    }
    const int wpr = W / rows;
    k_msm_recode_batch<<<dim3((unsigned)((n_pad + MSM_BLOCK - 1) / MSM_BLOCK), K), MSM_BLOCK, 0, ctx->stream>>>(
        b, n_main, extra_pos, n_extra, n_pad, digits, digits_per_commitment, c, W, wpr, (size_t)rows * n_pad, modulus,
        ctx->d_status);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

int msm_recode_wide_batch(vmpc_ctx *ctx, const void *const *scalars, size_t n_main, const void *const *extra_scalars,
                          int K, size_t extra_pos, size_t n_extra, size_t row_stride, int32_t *digits32,
                          size_t digits_per_commitment, const msm_modulus &modulus) {
    vmpc_stage_scope s(ctx, "msm_recode");
    msm_recode_batch b;
    memset(&b, 0, sizeof b);
    for (int k = 0; k < K; k++) {
        b.sc[k] = (const uint32_t *)scalars[k];
        b.sc_extra[k] = extra_scalars ? (const uint32_t *)extra_scalars[k] : nullptr;
    }
    k_msm_recode_wide<<<dim3((unsigned)((row_stride + MSM_BLOCK - 1) / MSM_BLOCK), K), MSM_BLOCK, 0, ctx->stream>>>(
        b, n_main, extra_pos, n_extra, row_stride, digits32, digits_per_commitment, modulus, ctx->d_status);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

// hist1 -> scan -> part1 -> fine -> plan over digits already in w.digits
template <bool FINE_IN_ENTRY, int CAP>
static int msm_launch_fine(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w) {
    hipStream_t st = ctx->stream;
    const unsigned grid = (unsigned)p.NC * (unsigned)p.W;
    const size_t dyn = CAP > SORT_FINE_CAP ? (size_t)CAP * 4 : 0;
    if (dyn && !ctx->sort_wide_ready) {
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_sort_fine<FINE_IN_ENTRY, CAP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_sort_fine_big<FINE_IN_ENTRY, CAP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        ctx->sort_wide_ready = true;
    }
    k_sort_fine<FINE_IN_ENTRY, CAP><<<grid, SORT_BLOCK, dyn, st>>>(
        w.stage1, w.hist1, p.NC, p.W, p.LB, p.top_row, p.period, p.LB_top, p.J, p.idx_bits, p.chunks_per_row,
        (uint32_t)p.row_stride, p.nb1, w.digits, p.n_pad, w.counts, w.starts, w.sorted, p.seg_shift, p.balanced, w.nseg,
        w.block_hist, w.heavy_list, w.seg_starts, w.ctrl);
    k_sort_fine_big<FINE_IN_ENTRY, CAP><<<grid, SORT_BLOCK, dyn, st>>>(
        w.stage1, w.hist1, p.NC, p.W, p.LB, p.top_row, p.period, p.LB_top, p.J, p.idx_bits, p.chunks_per_row,
        (uint32_t)p.row_stride, p.nb1, w.digits, p.n_pad, w.starts, w.sorted, w.ctrl);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

int msm_sort_digits(vmpc_ctx *ctx, const msm_plan &p, msm_ws &w) {
    hipStream_t st = ctx->stream;
    if (p.wide && (!p.fine_in_entry || p.chunks_per_row <= 0 || p.J != p.chunks_per_row * MSM_WIDE_ROWS ||
                   p.n_pad != (size_t)p.J * SORT_T || p.row_stride != (size_t)p.chunks_per_row * SORT_T))
        return VMPC_E_INVAL;
    // wide window: chunks are dealt to the XCDs in contiguous ranges (sort_xcd_chunk): 8 * ceil(J / 8) workgroups
    static const int xcd = getenv("VMPC_WIDE_XCD") ? atoi(getenv("VMPC_WIDE_XCD")) : 1;      // (A/B knob)
    const dim3 chunk_grid(p.wide ? 8 * ((p.J + 7) / 8) : p.J, p.W);
    {
        vmpc_stage_scope s(ctx, "msm_hist");
        if (p.wide)
            k_sort_hist1<true><<<chunk_grid, SORT_BLOCK, (size_t)p.NC * 4, st>>>(w.digits, p.n_pad, p.NC, p.LB, p.top_row,
                                                                                p.period, p.LB_top, p.J, xcd, w.hist1, w.ctrl);
        else
            k_sort_hist1<false><<<chunk_grid, SORT_BLOCK, (size_t)p.NC * 4, st>>>(w.digits, p.n_pad, p.NC, p.LB, p.top_row,
                                                                                 p.period, p.LB_top, p.J, 0, w.hist1, w.ctrl);
        VMPC_KERNEL_CHECK();
        VMPC_CHECK((vmpc_exclusive_scan<uint32_t, uint32_t>(st, w.hist1, w.hist1, w.hist1_n, w.scan_ws,
                                                            w.hist1 + w.hist1_n)));    // [H] = #entries
    }
    {
        vmpc_stage_scope s(ctx, "msm_part");
        const size_t lds_bytes = ((size_t)3 * p.NC + 16 + (p.wide ? 2 : 1) * SORT_T) * 4;
        if (p.wide) {
            if (lds_bytes > 48 * 1024)
                VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_sort_part1<true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            k_sort_part1<true><<<chunk_grid, SORT_BLOCK, lds_bytes, st>>>(
                w.digits, p.n_pad, p.NC, p.LB, p.top_row, p.period, p.LB_top, p.J, p.idx_bits, p.chunks_per_row, xcd,
                w.hist1, w.stage1);
        } else {
            if (lds_bytes > 48 * 1024)
                VMPC_HIP_CHECK(hipFuncSetAttribute((const void *)k_sort_part1<false>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            k_sort_part1<false><<<chunk_grid, SORT_BLOCK, lds_bytes, st>>>(
                w.digits, p.n_pad, p.NC, p.LB, p.top_row, p.period, p.LB_top, p.J, p.fine_in_entry ? p.idx_bits : 31, 0, 0,
                w.hist1, w.stage1);
        }
        VMPC_KERNEL_CHECK();
    }
    {
        vmpc_stage_scope s(ctx, "msm_sort");
        if (p.wide) VMPC_CHECK((msm_launch_fine<true, 16384>(ctx, p, w)));
        else if (p.fine_in_entry) VMPC_CHECK((msm_launch_fine<true, SORT_FINE_CAP>(ctx, p, w)));
        else VMPC_CHECK((msm_launch_fine<false, SORT_FINE_CAP>(ctx, p, w)));
    }
    {
        vmpc_stage_scope s(ctx, "msm_plan");
        if (p.W <= 8 && (size_t)p.W * p.NC <= 1024) {
            // (the windows one after the other in ONE workgroup: only when that is short - three wide-window
            // commitments of 1024 bins each took 81 us this way, 27 us side by side)
            k_msm_ranks_classes<<<1, 1024, 0, st>>>(w.block_hist, p.NC, p.W, w.ctrl, w.block_base);
            VMPC_KERNEL_CHECK();
        } else {
            k_msm_ranks<<<p.W, 1024, 0, st>>>(w.block_hist, p.NC, p.W, w.ctrl);
            VMPC_KERNEL_CHECK();
            k_msm_classes<<<1, 1024, 0, st>>>(w.ctrl, p.W, w.block_base);
            VMPC_KERNEL_CHECK();
        }
        k_msm_plan2<<<w.plan_blocks, 512, 0, st>>>(w.counts, p.NC, p.W, 1 << p.LB, p.top_row, p.period, 1 << p.LB_top,
                                                  p.nb1, p.seg_shift, p.balanced, w.block_base, w.block_hist, w.tasks);
        VMPC_KERNEL_CHECK();
    }
    return VMPC_OK;
}
