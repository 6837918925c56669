// Device-resident round context of the Protocol-4 prover ("vmpc_ctx_create(generators...),
// vmpc_ctx_round(ctx, c[32], out_A[64], out_B[64])").
//
// One halving round of verifiable_mpc/ac20/compressed_pivot.py:29-86 is
//     A_i = commit(z_l, L~(0 || z_l), g_r, k)   B_i = commit(z_r, L~(z_r || 0), g_l, k)      (:41-42)
//     c   = hash(...)                                                                     (host)
//     z'  = z_l + c z_r     L' = c L_l + L_r     (g' = fold, or - here - challenge products)  (:64-76)
// This context keeps z_hat, L~ and the per-generator challenge products in HBM across all log N rounds and runs
// a round as ONE C call: fold z and L~ with the previous challenge, the two inner products straight into the
// exponent slots of k, the block of commitment scalars (k_fr_tail_scalars_inc), and A_i, B_i as one batched
// pass over the tabulated CRS (vmpc_msm_table_batch_dev).  Only the 2 x 64 bytes of A_i, B_i come back and the
// 32-byte challenge goes in; the Fiat-Shamir hash stays with the caller (it defines the transcript).
// The generators are never folded: round i commits to the UNFOLDED CRS with the pending challenge products in the
// scalars (same group elements as the fold - tests/test_gpu_protocol.py compares with the round-by-round form).
//
// After VMPC_P4_JUMP rounds (default 5) on a large CRS the pending challenge products ARE applied: one pass of
// vmpc_msm_table_fold_dev (fold_jump.hip) produces the 2^5-times shorter folded vector, which gets its own small
// table, and the remaining rounds run on that - a round over 2^15 generators is bound by launch latency
// (0.4 ms), one over the unfolded 2^20 by its 16 M bucket additions (1.7 ms).
//
// Sharded form (vmpc_p4_create_sharded): g_hat is cut into `world` CONTIGUOUS blocks, one per rank; this rank's table
// holds block `rank`.  z_hat and L~ are replicated (their folds are 2N scalar operations over the whole proof), the
// per-generator scalars (challenge products, v_a, v_b) exist for the rank's block only, A_i and B_i are partial sums
// followed by ONE exchange per round (vmpc_comm_points_allsum_dev: all-gather of 2 x 128 bytes per rank on the
// context's stream + rank-ordered add), after which every rank holds the same bits and derives the same challenge.
// The jump stays local: index j + b N/2^k of the k-round fold lies in block (j + b N/2^k) / (N/world), so with
// 2^k >= world a block holds 2^k / world whole strides and the rank folds ITS strides into a partial vector
//       part_r[j] = sum_{b in the rank's strides} s_b g[j + b N/2^k],       g^(k)[j] = sum_r part_r[j];
// the parts are never added up: later rounds commit to part_r with the full (short, replicated) scalar vector and
// the exchange sums the partial commitments as before.  No generator ever crosses a link.
//
// The short end.  Once the vector is down to 2^11 generators (VMPC_P4_DIRECT_LOG2) a commitment skips the bucket
// method: with the folded vector's 16-row table a term v * G is 16 independent 16-bit products d_r * (2^(16 r) G), so one
// lane per (generator, row) runs a 16-step double-and-add on its table entry and a workgroup tree adds them up
// (k_p4_direct) - 0.15 ms per round where the sort -> bucket -> reduce -> recombine pipeline costs 0.3 ms of
// latency whatever the size (N = 2^10: 3.4 -> 2.7 ms per proof).  Folding a longer vector down to that size just to
// get there does not pay (p4_next_jump), so this serves vectors that are short to begin with.
//
// Built entirely on the public C-ABI of this library (include/vmpc.h) plus the host-callable field code.
#include <stdlib.h>

#include <array>
#include <chrono>
#include <thread>
#include <vector>

#include "common.h"
#include "fe25519.h"
#include "fr.h"
#include "ge25519.h"
#include "ptio.h"
#include "fe51_host.h"


// ---- the scalar side of one round in two launches ------------------------------------------------
#define P4_BLOCK 256
#define P4_MAX_GRID 1024

struct p4_scalar {
    uint32_t v[8];
};

__device__ __forceinline__ fr p4_ld(const uint32_t *src) {
    const uint4 *q = reinterpret_cast<const uint4 *>(src);
    const uint4 a = q[0], b = q[1];
    fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void p4_st(uint32_t *dst, const fr &a) {
    uint4 *q = reinterpret_cast<uint4 *>(dst);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// Fold with the previous challenge (z' = z_l + c z_r, L' = c L_l + L_r; compressed_pivot.py:70-76) and, in the
// same pass, the inner products of the NEW vectors that are the exponents of k in the next A_i, B_i (:41-42):
//   gamma_a = <L'[h:], z'[:h]>,   gamma_b = <L'[:h], z'[h:]>,   h = len(z') / 2.
// Lane i < h produces elements i and h + i of both new vectors.  fold = 0: first round, nothing to fold.
__global__ void __launch_bounds__(P4_BLOCK)
k_p4_fold_dots(p4_scalar c, const uint32_t *__restrict__ c_mem, int fold, const uint32_t *__restrict__ z_in,
               const uint32_t *__restrict__ L_in,
               size_t m_out, uint32_t *__restrict__ z_out, uint32_t *__restrict__ L_out,
               uint32_t *__restrict__ partials /* [2][gridDim.x] */) {
    __shared__ uint32_t lds[2 * P4_BLOCK * 8];
    fr cc;                                  // c_mem: a round queued before its challenge existed (vmpc_p4_run_compact)
#pragma unroll
    for (int i = 0; i < 8; i++) cc.v[i] = c_mem ? c_mem[i] : c.v[i];
    const size_t h = m_out / 2;
    fr pa = fr_zero(), pb = fr_zero();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < h; i += (size_t)gridDim.x * blockDim.x) {
        fr z0, z1, l0, l1;
        if (fold) {
            z0 = fr_add(p4_ld(z_in + 8 * i), fr_mul(cc, p4_ld(z_in + 8 * (m_out + i))));
            z1 = fr_add(p4_ld(z_in + 8 * (h + i)), fr_mul(cc, p4_ld(z_in + 8 * (m_out + h + i))));
            l0 = fr_add(fr_mul(cc, p4_ld(L_in + 8 * i)), p4_ld(L_in + 8 * (m_out + i)));
            l1 = fr_add(fr_mul(cc, p4_ld(L_in + 8 * (h + i))), p4_ld(L_in + 8 * (m_out + h + i)));
            p4_st(z_out + 8 * i, z0);
            p4_st(z_out + 8 * (h + i), z1);
            p4_st(L_out + 8 * i, l0);
            p4_st(L_out + 8 * (h + i), l1);
        } else {
            z0 = p4_ld(z_in + 8 * i);
            z1 = p4_ld(z_in + 8 * (h + i));
            l0 = p4_ld(L_in + 8 * i);
            l1 = p4_ld(L_in + 8 * (h + i));
        }
        pa = fr_add(pa, fr_mul(l1, z0));
        pb = fr_add(pb, fr_mul(l0, z1));
    }
    p4_st(lds + 8 * threadIdx.x, pa);
    p4_st(lds + 8 * (P4_BLOCK + threadIdx.x), pb);
    __syncthreads();
    for (int stride = P4_BLOCK / 2; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride) {
            p4_st(lds + 8 * threadIdx.x, fr_add(p4_ld(lds + 8 * threadIdx.x), p4_ld(lds + 8 * (threadIdx.x + stride))));
            p4_st(lds + 8 * (P4_BLOCK + threadIdx.x), fr_add(p4_ld(lds + 8 * (P4_BLOCK + threadIdx.x)),
                                                             p4_ld(lds + 8 * (P4_BLOCK + threadIdx.x + stride))));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p4_st(partials + 8 * blockIdx.x, p4_ld(lds));
        p4_st(partials + 8 * (gridDim.x + blockIdx.x), p4_ld(lds + 8 * P4_BLOCK));
    }
}

// The extras' scalars of A_i (block 0) and B_i (block 1): zero everywhere, the tail of g_hat (h) from the
// commitment scalars, the sum of the inner-product partials at k's slot.
__global__ void __launch_bounds__(P4_BLOCK)
k_p4_extras(const uint32_t *__restrict__ partials, int n_partials, const uint32_t *__restrict__ va,
            const uint32_t *__restrict__ vb, size_t table_n, int h_slots, int k_slot, int n_extra,
            uint32_t *__restrict__ ex_a, uint32_t *__restrict__ ex_b) {
    __shared__ uint32_t lds[P4_BLOCK * 8];
    const uint32_t *part = partials + 8 * (size_t)blockIdx.x * n_partials;
    const uint32_t *v = blockIdx.x ? vb : va;
    uint32_t *ex = blockIdx.x ? ex_b : ex_a;
    fr acc = fr_zero();
    for (int i = threadIdx.x; i < n_partials; i += P4_BLOCK) acc = fr_add(acc, p4_ld(part + 8 * i));
    p4_st(lds + 8 * threadIdx.x, acc);
    __syncthreads();
    for (int stride = P4_BLOCK / 2; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride)
            p4_st(lds + 8 * threadIdx.x, fr_add(p4_ld(lds + 8 * threadIdx.x), p4_ld(lds + 8 * (threadIdx.x + stride))));
        __syncthreads();
    }
    for (int s = threadIdx.x; s < n_extra; s += P4_BLOCK) {
        fr e = fr_zero();       // k_slot < 0: this rank does not add the k term (sharded prover: rank 0 does)
        if (s < h_slots) e = p4_ld(v + 8 * (table_n + s));
        else if (s == k_slot) e = p4_ld(lds);
        p4_st(ex + 8 * s, e);
    }
}

// ---- commitments over a short tabulated vector without buckets ---------------------------------------------------
// A_i (workgroups [0, blocks_per)) and B_i ([blocks_per, 2 blocks_per)) over the table's n_cols = generators + extras
// columns: lane t of a commitment handles row r = t / n_cols of column t % n_cols, i.e. the digit
// d = bits [r dbits, (r+1) dbits) of the column's scalar against the table entry 2^(r dbits) * G (dbits = 256 / rows),
// by a dbits-step double-and-add; the workgroup then adds its 256 results in an LDS tree.
#define P4D_BLOCK 256

__device__ __forceinline__ void p4d_tree(uint32_t *lds, const ge_ext &mine, uint32_t *dst, bool packed) {
    ext_st(lds + EXT_WORDS * threadIdx.x, mine);
    __syncthreads();
    for (uint32_t stride = P4D_BLOCK / 2; stride >= 1; stride >>= 1) {
        if (threadIdx.x < stride)
            ext_st(lds + EXT_WORDS * threadIdx.x,
                   ge_add(ext_ld(lds + EXT_WORDS * threadIdx.x), ext_ld(lds + EXT_WORDS * (threadIdx.x + stride))));
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (packed) ext_st8(dst, ext_ld(lds));
        else ext_st(dst, ext_ld(lds));
    }
}

__global__ void __launch_bounds__(P4D_BLOCK)
k_p4_direct(const uint32_t *__restrict__ table, size_t stride, int rows, size_t table_n, size_t n_cols,
            const uint32_t *__restrict__ va, const uint32_t *__restrict__ vb, const uint32_t *__restrict__ ex_a,
            const uint32_t *__restrict__ ex_b, unsigned blocks_per, uint32_t *__restrict__ partial) {
    __shared__ uint32_t lds[P4D_BLOCK * EXT_WORDS];
    const unsigned which = blockIdx.x / blocks_per;
    const size_t t = (size_t)(blockIdx.x % blocks_per) * P4D_BLOCK + threadIdx.x;
    const int dbits = 256 / rows;                      // 16 or 32
    ge_ext acc = ge_ext_identity();
    if (t < n_cols * (size_t)rows) {
        const int r = (int)(t / n_cols);
        const size_t col = t % n_cols;
        const uint32_t *sc = col < table_n ? (which ? vb : va) + 8 * col : (which ? ex_b : ex_a) + 8 * (col - table_n);
        const uint32_t d = dbits == 32 ? sc[r] : (sc[r >> 1] >> (16 * (r & 1))) & 0xffffu;
        if (d) {
            const ge_niels q = niels_ld_line(table + NIELS_WORDS * ((size_t)r * stride + col));
            for (int b = dbits - 1; b >= 0; b--) {
                acc = ge_dbl(acc);
                if ((d >> b) & 1u) acc = ge_madd(acc, q);
            }
        }
    }
    p4d_tree(lds, acc, partial + EXT_WORDS * (size_t)blockIdx.x, false);
}

// the blocks_per partial sums of each commitment -> its sum, 128 bytes X || Y || Z || T at out + 128 * blockIdx.x
__global__ void __launch_bounds__(P4D_BLOCK)
k_p4_direct_sum(const uint32_t *__restrict__ partial, unsigned blocks_per, uint32_t *__restrict__ out_ext,
                uint32_t *done_counter, uint32_t *done_flag, uint32_t done_seq) {
    __shared__ uint32_t lds[P4D_BLOCK * EXT_WORDS];
    const uint32_t *src = partial + EXT_WORDS * (size_t)blockIdx.x * blocks_per;
    ge_ext acc = ge_ext_identity();
    for (unsigned j = threadIdx.x; j < blocks_per; j += P4D_BLOCK) acc = ge_add(acc, ext_ld(src + EXT_WORDS * (size_t)j));
    p4d_tree(lds, acc, out_ext + 32 * blockIdx.x, true);
    if (done_flag && threadIdx.x == 0) vmpc_publish_done(done_counter, done_flag, done_seq);
}

int vmpc_fr_check_dev(vmpc_ctx *ctx, const void *v, size_t n);      // frvec.hip: bumps the status word, no sync
int vmpc_fr_tail_scalars_block_mem(vmpc_ctx *ctx, const uint8_t newest_challenge[32], const uint32_t *challenge_mem,
                                   int t, int log2_m0, const void *z, size_t j0, size_t count, void *products,
                                   void *out_a, void *out_b);       // frvec.hip

int vmpc_table_fold_table_with_block(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                     size_t n_cols, int k, const uint8_t *scalars, size_t n_extra, int out_rows,
                                     const void *extras_block, void *out_table);   // fold_jump.hip

struct vmpc_p4 {
    vmpc_ctx *ctx;
    const void *table;
    size_t table_n, table_extra;
    int rows, k_slot, h_slots;
    int log2_n, round, committed;     // log2_n: of the vector the table holds (the base of the pending products)
    int total_rounds;
    size_t m;                         // current length of z_hat / L~
    char *z[2], *L[2];                // ping-pong
    char *products, *va, *vb, *ex_a, *ex_b, *out, *partials;
    int cur;
    unsigned dots_grid;               // workgroups of the last k_p4_fold_dots = partial sums per inner product
    // fold jump: challenges not yet applied to the generators, the folded vector and its table
    std::vector<std::array<uint8_t, 32>> pending;
    int jump_k;
    size_t jump_min;
    char *k_aff;
    // the folds this proof will make, fixed at creation (p4_next_jump): their tables come out of the arena
    struct jump_slot { int k; size_t m_out, bytes; char *table; };
    std::vector<jump_slot> jumps;
    size_t jumps_done;
    int direct_log2;                  // commitments over <= 2^direct_log2 generators skip the bucket method (0: never)
    char *arena;
    bool arena_pooled;
    std::vector<char *> extra;        // buffers of second and later jumps
    uint8_t k_host[64];
    // sharded form: this rank's block of g_hat is [block_lo, block_lo + block_n) until the generators are folded,
    // afterwards every rank holds a full-length partial vector (block_lo = 0)
    vmpc_comm *comm;
    int world, rank;
    size_t block_lo, block_n;
    char *mine, *gathered;            // 2 partial points of this rank; world x 2 gathered ones
    bool poisoned;                    // a call failed after the fold state advanced: only destroy is valid
    bool in_flight;                   // vmpc_p4_round_begin without its vmpc_p4_round_end yet
    bool lazy_fold;                   // a due fold waits for vmpc_p4_prefold (or the NEXT round), see vmpc_p4_create_opts
    int products_t;                   // challenges the per-generator products hold (k_fr_tail_scalars_inc), -1: none
    const void *table0;               // the caller's table (the unfolded CRS / block)
    // a second table over the SAME generators and extras for the pair commitments of the rounds before the fold
    // (vmpc_p4_set_commit_table: the 13-row wide-window table - 13 mixed additions per term instead of 16); the fold
    // itself keeps reading `table0`, whose rows are spaced 256 / rows bits
    const void *commit_table;
    int commit_rows;
    size_t commit_min_cols;
};

// All device buffers of a context are carved from one arena that stays with the vmpc_ctx between proofs
// (ctx->p4_pool): a prover that runs proof after proof pays no hipMalloc / hipFree (which synchronises the
// device) per proof.  A second live context on the same vmpc_ctx gets a private arena.
static void p4_release(vmpc_p4 *p) {
    if (!p) return;
    for (char *b : p->extra) (void)hipFree(b);
    if (p->arena) {
        if (p->arena_pooled) p->ctx->p4_pool_busy = false;
        else (void)hipFree(p->arena);
    }
    delete p;
}

// rows of the folded vector's table: as PointVector.precompute picks them (1 GiB at most)
static int p4_jump_rows(size_t m_out) {
    int rows = 16;
    while (rows > 1 && (size_t)rows * 128 * (m_out + 1) > ((size_t)1 << 30)) rows /= 2;
    return rows;
}

static size_t p4_align(size_t b) { return (b + 255) & ~(size_t)255; }

static int p4_log2(size_t v) {
    int l = 0;
    while (((size_t)1 << l) < v) l++;
    return l;
}

// How many rounds' challenges the next fold of the generators applies, for a vector of 2^log2_n generators of which
// this rank's table holds block_n (spread over `spread` ranks before the first fold, 1 afterwards); 0 = no further
// fold.  Two rules: a large block is folded after jump_k rounds (64 additions per generator whatever k, and the
// rounds before it cost a full commitment each); a vector of at most 2^17 generators is folded once more (or for
// the first time), straight down to the 2^direct_log2 generators the bucket-free commitment handles.  Sharded,
// a block must hold at least two strides of the fold: 2^k >= 2 * spread.
static int p4_next_jump(int jump_k, size_t jump_min, int direct_log2, bool fold_to_direct, int log2_n, size_t block_n,
                        size_t spread) {
    if (!jump_k) return 0;
    if (block_n >= jump_min && log2_n - jump_k >= 2 && ((size_t)1 << jump_k) >= 2 * spread) return jump_k;
    // (the second rule is off by default: the fold itself - a chain of rows x 2^k table entries per lane plus a
    // 240-doubling table build, 1.2 ms at any size - costs what the cheaper rounds save: 18.9 ms either way at
    // N = 2^20, 7.2 against 6.6 ms at 2^16; VMPC_P4_FOLD_TO_DIRECT=1 turns it on)
    const int k = log2_n - direct_log2;
    if (fold_to_direct && direct_log2 && k >= 1 && k <= 6 && ((size_t)1 << k) >= 2 * spread) return k;
    return 0;
}

// table: fixed-base table over table_n generators followed by table_extra extras, of which extras
// 0 .. h_slots-1 are the tail of g_hat (g_hat = g || h: h_slots = 1) and extra `k_slot` is k.  The table's
// generators (and tail) are block `rank` of g_hat's `world` equal blocks (comm = NULL: the whole of g_hat).
// z_hat, L_tilde: N = world * (table_n + h_slots) scalars each (device, canonical residues), N a power of two >= 4.
static int p4_create(vmpc_ctx *ctx, vmpc_comm *comm, const void *table, size_t table_n, size_t table_extra, int rows,
                     int h_slots, int k_slot, const uint8_t k_affine[64], const void *z_hat, const void *L_tilde,
                     vmpc_p4 **out, int jump_k_req = -1) {
    if (!ctx || !table || !z_hat || !L_tilde || !out || !k_affine || h_slots < 0 || k_slot < h_slots ||
        (size_t)k_slot >= table_extra || !(rows == 1 || rows == 2 || rows == 4 || rows == 8 || rows == 16))
        return VMPC_E_INVAL;       // (rows = 13, the wide-window table, serves commitments only)
    int world = 1, rank = 0;
    if (comm) VMPC_CHECK(vmpc_comm_info(comm, &world, &rank, nullptr));
    // the tail slots (h) belong to the last block only; a sharded CRS keeps h as that block's last generator
    if (world < 1 || (world & (world - 1)) || (world > 1 && h_slots != 0)) return VMPC_E_INVAL;
    const size_t block_n = table_n + (size_t)h_slots;
    const size_t N = block_n * (size_t)world;
    if (N < 4 || (N & (N - 1)) || block_n < 1) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_p4 *p = new vmpc_p4();
    p->z[0] = p->z[1] = p->L[0] = p->L[1] = p->products = p->va = p->vb = p->ex_a = p->ex_b = p->out = nullptr;
    p->k_aff = p->partials = p->mine = p->gathered = nullptr;
    p->arena = nullptr;
    p->commit_table = nullptr;
    p->commit_rows = 0;
    p->commit_min_cols = 0;
    p->arena_pooled = false;
    p->poisoned = false;
    p->in_flight = false;
    p->lazy_fold = false;
    p->products_t = -1;
    p->dots_grid = 0;
    p->round = p->committed = p->cur = p->log2_n = 0;
    p->jump_k = 5;                               // VMPC_P4_JUMP=0 keeps every round on the unfolded CRS
    if (const char *e = getenv("VMPC_P4_JUMP")) p->jump_k = atoi(e);
    if (jump_k_req >= 0) p->jump_k = jump_k_req;             // (vmpc_p4_create_opts)
    if (p->jump_k < 0 || p->jump_k > 6) p->jump_k = 0;
    p->jump_min = (size_t)1 << 18;
    if (const char *e = getenv("VMPC_P4_JUMP_MIN_LOG2")) p->jump_min = (size_t)1 << atoi(e);
    p->direct_log2 = 11;                         // VMPC_P4_DIRECT_LOG2=0: every commitment by the bucket method
    if (const char *e = getenv("VMPC_P4_DIRECT_LOG2")) p->direct_log2 = atoi(e);
    if (p->direct_log2 < 2 || p->direct_log2 > 14) p->direct_log2 = 0;
    p->jumps_done = 0;
    p->ctx = ctx;
    p->comm = comm;
    p->world = world;
    p->rank = rank;
    p->block_lo = (size_t)rank * block_n;
    p->block_n = block_n;
    p->table = p->table0 = table;
    p->table_n = table_n;
    p->table_extra = table_extra;
    p->rows = rows;
    p->h_slots = h_slots;
    p->k_slot = k_slot;
    p->m = N;
    p->log2_n = p4_log2(N);
    p->total_rounds = p->log2_n - 1;
    // the folds of this proof and the tables they leave, all from the arena
    {
        int l2 = p->log2_n;
        size_t bn = block_n, spread = (size_t)world;
        const char *f2d = vmpc_getenv_experimental("VMPC_P4_FOLD_TO_DIRECT");
        while (int k = p4_next_jump(p->jump_k, p->jump_min, p->direct_log2, f2d && atoi(f2d) != 0, l2, bn, spread)) {
            vmpc_p4::jump_slot js;
            js.k = k;
            js.m_out = ((size_t)1 << l2) >> k;
            js.table = nullptr;
            if (vmpc_msm_table_bytes(js.m_out, 1, p4_jump_rows(js.m_out), &js.bytes) != VMPC_OK) {
                delete p;
                return VMPC_E_INVAL;
            }
            p->jumps.push_back(js);
            l2 -= k;
            bn = js.m_out;
            spread = 1;
        }
    }
    const size_t per_gen = 32 * block_n;          // products, v_a, v_b: one entry per local column (fewer after a jump)
    std::vector<size_t> sizes = {32 * N, 32 * N, 32 * N, 32 * N, per_gen, per_gen, per_gen, 32 * table_extra + 32,
                                 32 * table_extra + 32, 256, (size_t)2 * P4_MAX_GRID * 32, 64, 256, (size_t)256 * world};
    std::vector<char **> slots = {&p->z[0], &p->z[1], &p->L[0], &p->L[1], &p->products, &p->va, &p->vb, &p->ex_a,
                                  &p->ex_b, &p->out, &p->partials, &p->k_aff, &p->mine, &p->gathered};
    for (auto &js : p->jumps) {
        sizes.push_back(js.bytes);
        slots.push_back(&js.table);
    }
    size_t total = 0;
    for (size_t b : sizes) total += p4_align(b);
    if (!ctx->p4_pool_busy) {
        if (ctx->p4_pool_bytes < total) {
            VMPC_IGNORE(hipStreamSynchronize(ctx->stream));
            if (ctx->p4_pool) VMPC_IGNORE(hipFree(ctx->p4_pool));
            ctx->p4_pool = nullptr;
            ctx->p4_pool_bytes = 0;
            if (hipMalloc(&ctx->p4_pool, total) != hipSuccess) {
                delete p;
                return VMPC_E_NOMEM;
            }
            ctx->p4_pool_bytes = total;
        }
        p->arena = (char *)ctx->p4_pool;
        p->arena_pooled = true;
        ctx->p4_pool_busy = true;
    } else {
        if (hipMalloc((void **)&p->arena, total) != hipSuccess) {
            delete p;
            return VMPC_E_NOMEM;
        }
        p->arena_pooled = false;
    }
    {
        size_t off = 0;
        for (size_t i = 0; i < sizes.size(); i++) {
            *slots[i] = sizes[i] ? p->arena + off : nullptr;
            off += p4_align(sizes[i]);
        }
    }
    memcpy(p->k_host, k_affine, 64);
    int rc = vmpc_memcpy_h2d(ctx, p->k_aff, k_affine, 64);
    if (rc == VMPC_OK) rc = vmpc_memcpy_d2d(ctx, p->z[0], z_hat, 32 * N);
    if (rc == VMPC_OK) rc = vmpc_memcpy_d2d(ctx, p->L[0], L_tilde, 32 * N);
    // canonical residues?  Counted into the context's status word here, for EVERY path the rounds may take (the
    // bucket-free commitments of short vectors never recode, so they would not notice); the first round's
    // vmpc_ctx_sync reports it as VMPC_E_NONCANON.  Two streaming reads of 32 N bytes.
    if (rc == VMPC_OK) rc = vmpc_fr_check_dev(ctx, p->z[0], N);
    if (rc == VMPC_OK) rc = vmpc_fr_check_dev(ctx, p->L[0], N);
    if (rc != VMPC_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        p4_release(p);
        return rc;
    }
    *out = p;
    return VMPC_OK;
}

extern "C" int vmpc_p4_set_commit_table(vmpc_p4 *p, const void *table, int rows) {
    if (!p || p->poisoned || (table && rows != 13 && !(rows == 1 || rows == 2 || rows == 4 || rows == 8 || rows == 16)))
        return VMPC_E_INVAL;
    p->commit_table = table;
    p->commit_rows = table ? rows : 0;
    p->commit_min_cols = (size_t)1 << 17;
    if (const char *e = getenv("VMPC_P4_COMMIT_TABLE_MIN_LOG2")) p->commit_min_cols = atoi(e) > 0 ? (size_t)1 << atoi(e) : 0;   // (tests)
    return VMPC_OK;
}

extern "C" int vmpc_p4_create(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                              int h_slots, int k_slot, const uint8_t k_affine[64], const void *z_hat,
                              const void *L_tilde, vmpc_p4 **out) {
    return p4_create(ctx, nullptr, table, table_n, table_extra, rows, h_slots, k_slot, k_affine, z_hat, L_tilde, out);
}

// The same with the number of rounds before the generator fold chosen by the caller (0: never; < 0: the default), and
// with a LAZY fold: the round that is given the jump_k-th challenge still commits over the unfolded table (one more
// challenge product in its scalars) instead of folding first, and the fold is made by vmpc_p4_prefold - enqueued on the
// context's stream, nothing waited for - or, if the caller never asks, at the start of the round after.  For a caller
// with time between two rounds: the reference-transcript prover hashes megabytes of text on the host while the GPU is
// idle (compressed_pivot.py:51-59), and a 3.7-ms fold in front of a round's pair is 3.7 ms in front of that hash.
extern "C" int vmpc_p4_create_opts(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                   int h_slots, int k_slot, const uint8_t k_affine[64], const void *z_hat,
                                   const void *L_tilde, int jump_k, int lazy_fold, vmpc_p4 **out) {
    const int rc = p4_create(ctx, nullptr, table, table_n, table_extra, rows, h_slots, k_slot, k_affine, z_hat, L_tilde,
                             out, jump_k);
    if (rc == VMPC_OK) (*out)->lazy_fold = lazy_fold != 0;
    return rc;
}

static bool p4_jump_due(const vmpc_p4 *p);
static int p4_jump(vmpc_p4 *p);
extern "C" int vmpc_p4_prefold(vmpc_p4 *p) {
    // (not while a round is in flight: staging the fold's schedule may re-allocate the pinned block that round's
    // result is written to)
    if (!p || p->poisoned || p->in_flight) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(p->ctx->device));
    if (!p4_jump_due(p)) return VMPC_OK;
    const int rc = p4_jump(p);
    if (rc != VMPC_OK) p->poisoned = true;
    return rc;
}

extern "C" int vmpc_p4_create_sharded(vmpc_ctx *ctx, vmpc_comm *comm, const void *block_table, size_t block_n,
                                      size_t table_extra, int rows, int k_slot, const uint8_t k_affine[64],
                                      const void *z_hat, const void *L_tilde, vmpc_p4 **out) {
    if (!comm) return VMPC_E_INVAL;
    return p4_create(ctx, comm, block_table, block_n, table_extra, rows, 0, k_slot, k_affine, z_hat, L_tilde, out);
}

extern "C" int vmpc_p4_destroy(vmpc_p4 *p) {
    if (!p) return VMPC_E_INVAL;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    p4_release(p);
    return VMPC_OK;
}

// z' = z_l + c z_r, L' = c L_l + L_r with the challenge of the round just hashed (c = NULL: first round, no fold),
// and the two inner products of the resulting vectors as partial sums (k_p4_fold_dots)
// c_mem: the challenge is not known yet - the kernel reads it from there, and the entry of `pending` is a
// placeholder the caller fills in before the next fold of the generators
static int p4_fold_dots(vmpc_p4 *p, const uint8_t *c, const uint32_t *c_mem = nullptr) {
    static const uint8_t placeholder[32] = {0};
    p4_scalar cs;
    memset(&cs, 0, sizeof cs);
    if (c) {
        memcpy(cs.v, c, 32);
        if (fr_geq_l(cs.v)) return VMPC_E_NONCANON;
    } else if (c_mem) {
        c = placeholder;
    }
    const size_t m_out = c ? p->m / 2 : p->m, h = m_out / 2;
    const int nx = c ? p->cur ^ 1 : p->cur;
    size_t g = (h + P4_BLOCK - 1) / P4_BLOCK;
    if (g > P4_MAX_GRID) g = P4_MAX_GRID;
    if (g == 0) g = 1;
    vmpc_stage_scope s(p->ctx, "p4_fold_dots");
    k_p4_fold_dots<<<(unsigned)g, P4_BLOCK, 0, p->ctx->stream>>>(
        cs, c_mem, c ? 1 : 0, (const uint32_t *)p->z[p->cur], (const uint32_t *)p->L[p->cur], m_out, (uint32_t *)p->z[nx],
        (uint32_t *)p->L[nx], (uint32_t *)p->partials);
    VMPC_KERNEL_CHECK();
    p->dots_grid = (unsigned)g;
    if (c) {
        p->cur = nx;
        p->m = m_out;
        p->round++;
        std::array<uint8_t, 32> a;
        memcpy(a.data(), c, 32);
        p->pending.push_back(a);
    }
    return VMPC_OK;
}

// the next fold of the schedule is due: its challenges are all there
static bool p4_jump_due(const vmpc_p4 *p) {
    return p->jumps_done < p->jumps.size() && (int)p->pending.size() == p->jumps[p->jumps_done].k;
}

// Apply the pending challenges to the generators: g' = k folds of the table's vector, then a table for g' || k.
// Sharded: the rank folds the strides its block holds into a partial vector of the same length (see the top).
static int p4_jump(vmpc_p4 *p) {
    vmpc_ctx *ctx = p->ctx;
    const int k = (int)p->pending.size();
    const size_t N = (size_t)1 << p->log2_n, m_out = N >> k;
    const bool spread = p->block_n < N;                       // g_hat still in blocks over the ranks
    const int k_loc = k - (spread ? p4_log2((size_t)p->world) : 0);
    const size_t b0 = spread ? (size_t)p->rank << k_loc : 0;  // first stride of this rank's block
    // s_b = prod_i (c_i if bit (k - i) of b is 0), i = 1..k: the products k_fr_tail_scalars_inc keeps per generator
    std::vector<uint8_t> s((size_t)32 << k);
    for (int b = 0; b < (1 << k); b++) {
        fr v = fr_zero();
        v.v[0] = 1;
        for (int i = 0; i < k; i++)
            if (!((b >> (k - 1 - i)) & 1)) v = fr_mul(v, fr_load((const uint32_t *)p->pending[i].data()));
        fr_store((uint32_t *)(s.data() + 32 * b), v);
    }
    const uint8_t *s_loc = s.data() + 32 * b0;
    const int rows = p4_jump_rows(m_out);
    size_t bytes = 0;
    VMPC_CHECK(vmpc_msm_table_bytes(m_out, 1, rows, &bytes));
    if (p->jumps_done >= p->jumps.size() || p->jumps[p->jumps_done].m_out != m_out || p->jumps[p->jumps_done].bytes < bytes)
        return VMPC_E_INVAL;                                  // (the schedule fixed at creation)
    char *t = p->jumps[p->jumps_done].table;
    if (m_out % 8 == 0) {
        // k's columns of the new table: the same for every proof over this CRS
        if (!ctx->p4_kblock || ctx->p4_kblock_rows != rows || memcmp(ctx->p4_kblock_key, p->k_host, 64) != 0) {
            if (!ctx->p4_kblock && hipMalloc(&ctx->p4_kblock, (size_t)16 * 8 * 128) != hipSuccess) return VMPC_E_NOMEM;
            ctx->p4_kblock_rows = 0;
            VMPC_CHECK(vmpc_msm_table_build_dev(ctx, nullptr, 0, p->k_aff, 1, rows, ctx->p4_kblock));
            ctx->p4_kblock_rows = rows;
            memcpy(ctx->p4_kblock_key, p->k_host, 64);
        }
        VMPC_CHECK(vmpc_table_fold_table_with_block(ctx, p->table, p->table_n, p->table_extra, p->rows, p->block_n, k_loc,
                                                    s_loc, 1, rows, ctx->p4_kblock, t));
    } else {
        VMPC_CHECK(vmpc_msm_table_fold_table_dev(ctx, p->table, p->table_n, p->table_extra, p->rows, p->block_n, k_loc,
                                                 s_loc, p->k_aff, 1, rows, t));
    }
    p->table = t;
    p->table_n = m_out;
    p->table_extra = 1;
    p->rows = rows;
    p->h_slots = 0;
    p->k_slot = 0;
    p->log2_n -= k;
    p->block_lo = 0;                  // every rank now holds a full-length (partial) vector
    p->block_n = m_out;
    p->jumps_done++;
    p->pending.clear();
    p->products_t = -1;               // (the products were over the unfolded vector's positions)
    return VMPC_OK;
}

// A_i, B_i to affine on the host: one field inversion for the pair (1 / (Z_a Z_b)) in 51-bit limbs (fe51_host.h),
// ~3 us on a host core - cheaper than a 265-step single-lane chain on the GPU, and it sits between two rounds
static void p4_affine_pair(const uint8_t ext[256], uint8_t out_a[64], uint8_t out_b[64]) {
    fe51::el X[2], Y[2], Z[2];
    for (int i = 0; i < 2; i++) {
        X[i] = fe51::from_bytes(ext + 128 * i);
        Y[i] = fe51::from_bytes(ext + 128 * i + 32);
        Z[i] = fe51::from_bytes(ext + 128 * i + 64);
    }
    const fe51::el inv = fe51::inv(fe51::mul(Z[0], Z[1]));
    const fe51::el zi[2] = {fe51::mul(inv, Z[1]), fe51::mul(inv, Z[0])};
    uint8_t *out[2] = {out_a, out_b};
    for (int i = 0; i < 2; i++) {
        fe51::to_bytes(out[i], fe51::mul(X[i], zi[i]));
        fe51::to_bytes(out[i] + 32, fe51::mul(Y[i], zi[i]));
    }
}

// everything of a round that runs on the stream; c_mem as in p4_fold_dots (the fold before it was queued with it)
// Q' = A + c Q + c^2 B on the HOST (compressed_pivot.py:66: Q' = A * Q**c * B**(c**2)), affine x || y in and out.
// The reference transcript hashes Q' normalised, so any representative will do, and the host does the two 253-bit
// ladders in ~0.1 ms (fe51_host.h) - as a 3-term MSM on a side stream it was ~20 launches and three blocking
// uploads per round, all of them queued behind the round's generator fold.
extern "C" int vmpc_ed25519_fold_commitment_host(const uint8_t A[64], const uint8_t Q[64], const uint8_t B[64],
                                                 const uint8_t c[32], uint8_t out[64]) {
    if (!A || !Q || !B || !c || !out) return VMPC_E_INVAL;
    uint32_t cw[8];
    memcpy(cw, c, 32);
    if (fr_geq_l(cw)) return VMPC_E_NONCANON;
    const fr cf = fr_load(cw);
    uint8_t c2[32];
    fr_store((uint32_t *)c2, fr_mul(cf, cf));
    const fe51::el dd = fe51::d2();
    const fe51::pt a = fe51::pt_from_affine(A), q = fe51::pt_from_affine(Q), b = fe51::pt_from_affine(B);
    const fe51::pt r = fe51::pt_add(fe51::pt_add(a, fe51::pt_mul(q, c, dd), dd), fe51::pt_mul(b, c2, dd), dd);
    fe51::pt_to_affine(out, r);
    return VMPC_OK;
}

// out = sum_i s_i * P_i on the HOST, n <= 8 (affine in and out, canonical residues): Q_0 = A * P**c0 * k**(c1 (c0 y + t))
// of compressed_pivot.py:140, whose normalised value leads off the first round's pre-image - as Python big-int ladders it
// was 8 ms inside that round's hash call, for prover and verifier alike (round 6, scripts/ref_stall_probe.py)
extern "C" int vmpc_ed25519_lincomb_host(const uint8_t *points, const uint8_t *scalars, size_t n, uint8_t out[64]) {
    if (!points || !scalars || !out || n < 1 || n > 8) return VMPC_E_INVAL;
    const fe51::el dd = fe51::d2();
    fe51::pt acc;
    for (size_t i = 0; i < n; i++) {
        uint32_t cw[8];
        memcpy(cw, scalars + 32 * i, 32);
        if (fr_geq_l(cw)) return VMPC_E_NONCANON;
        const fe51::pt term = fe51::pt_mul(fe51::pt_from_affine(points + 64 * i), scalars + 32 * i, dd);
        acc = i ? fe51::pt_add(acc, term, dd) : term;
    }
    fe51::pt_to_affine(out, acc);
    return VMPC_OK;
}

static int p4_round_enqueue(vmpc_p4 *p, const uint32_t *c_mem) {
    vmpc_ctx *ctx = p->ctx;
    if (p4_jump_due(p) && !p->lazy_fold) {
        if (c_mem) return VMPC_E_INVAL;                     // a fold of the generators needs the challenges' values
        VMPC_CHECK(p4_jump(p));
    }
    const int t = (int)p->pending.size();                   // challenges the table's generators have not seen
    const char *z = p->z[p->cur];
    hipStream_t st = ctx->stream;
    // commitment scalars over the unfolded g_hat (this rank's block of it): challenge products x the (shifted)
    // witness halves
    static const uint8_t zero[32] = {0};
    if (t >= 1 && p->products_t != t - 1) {
        // the products are kept incrementally, one challenge per round; a lazy fold leaves the first round on the folded
        // table with one challenge pending and no products yet: start them (all ones) first
        if (t != 1) return VMPC_E_INVAL;
        VMPC_CHECK(vmpc_fr_tail_scalars_block_mem(ctx, zero, nullptr, 0, p->log2_n, z, p->block_lo, p->block_n, p->products,
                                                  p->va, p->vb));
    }
    VMPC_CHECK(vmpc_fr_tail_scalars_block_mem(ctx, t ? p->pending.back().data() : zero, c_mem, t, p->log2_n, z, p->block_lo,
                                              p->block_n, p->products, p->va, p->vb));
    p->products_t = t;
    // extras: the tail of g_hat (h) lives among them, and k with the inner products as exponents (rank 0 only:
    // the k term must enter the sum over the ranks once)
    {
        vmpc_stage_scope s(ctx, "p4_extras");
        k_p4_extras<<<2, P4_BLOCK, 0, st>>>((const uint32_t *)p->partials, (int)p->dots_grid, (const uint32_t *)p->va,
                                           (const uint32_t *)p->vb, p->table_n, p->h_slots,
                                           p->rank == 0 ? p->k_slot : -1, (int)p->table_extra, (uint32_t *)p->ex_a,
                                           (uint32_t *)p->ex_b);
        VMPC_KERNEL_CHECK();
    }
    const void *sc[2] = {p->va, p->vb}, *ex[2] = {p->ex_a, p->ex_b};
    // the last kernel writes the 2 x 128 bytes straight into pinned host memory: no copy command at all
    // (a pageable destination costs a staged copy, 20 us a round)
    VMPC_CHECK(vmpc_pinned_reserve(ctx, 0));
    void *pair_out = p->comm ? (void *)p->mine : ctx->pin_out_dev;       // partial sums go through the exchange
    if (p->direct_log2 && p->table_n <= ((size_t)1 << p->direct_log2) && p->rows >= 8) {
        // short vector: no buckets (k_p4_direct)
        const size_t n_cols = p->table_n + p->table_extra, stride = (n_cols + 7) & ~(size_t)7;
        const unsigned blocks_per = (unsigned)((n_cols * (size_t)p->rows + P4D_BLOCK - 1) / P4D_BLOCK);
        VMPC_CHECK(vmpc_ws_reserve(ctx, (size_t)2 * blocks_per * EXT_WORDS * 4 + 512));
        uint32_t *part = (uint32_t *)vmpc_ws_take(ctx, (size_t)2 * blocks_per * EXT_WORDS * 4);
        vmpc_stage_scope s(ctx, "p4_direct");
        k_p4_direct<<<2 * blocks_per, P4D_BLOCK, 0, st>>>((const uint32_t *)p->table, stride, p->rows, p->table_n, n_cols,
                                                          (const uint32_t *)p->va, (const uint32_t *)p->vb,
                                                          (const uint32_t *)p->ex_a, (const uint32_t *)p->ex_b, blocks_per,
                                                          part);
        VMPC_KERNEL_CHECK();
        k_p4_direct_sum<<<2, P4D_BLOCK, 0, st>>>(part, blocks_per, (uint32_t *)pair_out, ctx->d_status + VMPC_ST_WORDS,
                                                 ctx->done_flag_dev, ctx->done_seq);
        ctx->done_flag_dev = nullptr;
        VMPC_KERNEL_CHECK();
    } else {
        // v_a and v_b are each zero on half of their positions (z_l against g_r, z_r against g_l): tell the planner
        // (shift 2, not 1: the pair's 2 x 2^15 x period buckets hold ~128 entries each, and with 64-entry segments
        // the chip's 2^18 lanes get one full segment each plus a short second wave - 32-entry segments measured
        // 0.90 -> 0.75 ms for the bucket stage of a big round, scripts/prove_stages.py)
        ctx->plan_fill_shift = 2;
        if (const char *e = vmpc_getenv_experimental("VMPC_P4_FILL_SHIFT")) ctx->plan_fill_shift = atoi(e);
        // (narrower digits on a folded vector's table were measured in rounds 2-3 and lose at every size - DESIGN.md
        // section 10 - the knob is gone)
        // (under a communicator the fused short path stays off: its overflow answer - repeat on the general path - would
        // have to be agreed between the ranks)
        const int keep_short = ctx->short_path;
        if (p->comm) ctx->short_path = 0;
        // (only where the fused short path would not run anyway: tables of more than 2^17 columns)
        const bool over_commit_table = p->commit_table && p->table == p->table0 && !p->comm &&
                                       p->table_n + p->table_extra > p->commit_min_cols;
        const int rc = vmpc_msm_table_batch_dev(ctx, over_commit_table ? p->commit_table : p->table, p->table_n,
                                                p->table_extra, over_commit_table ? p->commit_rows : p->rows, sc,
                                                p->table_n, ex, 2, pair_out, nullptr);
        ctx->short_path = keep_short;
        ctx->plan_fill_shift = 0;
        VMPC_CHECK(rc);
    }
    // the round's one exchange: all-gather + rank-ordered add on the same stream, the result lands in the pinned block
    if (p->comm)
        VMPC_CHECK(vmpc_comm_points_allsum_dev(p->comm, ctx, p->mine, 2, p->gathered, ctx->pin_out_dev, nullptr));
    return VMPC_OK;
}

static int p4_round_collect(vmpc_p4 *p, uint8_t out_A[64], uint8_t out_B[64]) {
    vmpc_ctx *ctx = p->ctx;
    const uint8_t *ext = (const uint8_t *)ctx->pin_out;
    // The first round's synchronisation also fetches the device status words (a non-canonical scalar in the caller's
    // z_hat / L~ shows up in this round's recoding); later rounds only consume scalars this context produced, so
    // they just wait for the stream - vmpc_p4_finish checks the status once more at the end.
    if (p->committed == 0 || vmpc_getenv_experimental("VMPC_P4_FULL_SYNC")) VMPC_CHECK(vmpc_ctx_sync(ctx));
    else VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    p->committed++;
    p4_affine_pair(ext, out_A, out_B);
    return VMPC_OK;
}

static int p4_round_body(vmpc_p4 *p, uint8_t out_A[64], uint8_t out_B[64]) {
    VMPC_CHECK(p4_round_enqueue(p, nullptr));
    return p4_round_collect(p, out_A, out_B);
}

// lazy fold (vmpc_p4_create_opts): a fold that is still due when the next challenge arrives is made first - the
// schedule fixed at creation folds exactly jump_k challenges
static int p4_lazy_fold_now(vmpc_p4 *p) {
    if (!p->lazy_fold || !p4_jump_due(p)) return VMPC_OK;
    const int rc = p4_jump(p);
    if (rc != VMPC_OK) p->poisoned = true;
    return rc;
}

// One round: prev_challenge = the challenge derived from the PREVIOUS call's A, B (NULL on the first call).
// Returns A_i, B_i as 64-byte affine points.  Synchronises the context's stream (the results are needed
// for the next hash).  A failure after the fold has advanced the state leaves the context unusable (every
// further call returns VMPC_E_INVAL; destroy it).
extern "C" int vmpc_p4_round(vmpc_p4 *p, const uint8_t prev_challenge[32], uint8_t out_A[64], uint8_t out_B[64]) {
    // the first call has no challenge yet, every later one needs the previous round's; log2(N) - 1 rounds in all
    if (!p || p->poisoned || p->in_flight || !out_A || !out_B || (p->committed == 0) != (prev_challenge == nullptr) ||
        (prev_challenge && p->m / 2 < 4))
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(p->ctx->device));
    VMPC_CHECK(p4_lazy_fold_now(p));
    // fold with the previous challenge; exponents of k: L~(0 || z_l) and L~(z_r || 0) (compressed_pivot.py:41-42)
    VMPC_CHECK(p4_fold_dots(p, prev_challenge));      // rejects a non-canonical challenge before touching the state
    const int rc = p4_round_body(p, out_A, out_B);
    if (rc != VMPC_OK) p->poisoned = true;
    return rc;
}

// vmpc_p4_round in two halves: _begin enqueues the round on the context's stream and returns at once, _end waits for
// it and hands back A_i, B_i.  Between the two the context's stream (and its pinned result block) must be left alone:
// every other vmpc_p4_* call on this context returns VMPC_E_INVAL until _end has been called.
extern "C" int vmpc_p4_round_begin(vmpc_p4 *p, const uint8_t prev_challenge[32]) {
    if (!p || p->poisoned || p->in_flight || (p->committed == 0) != (prev_challenge == nullptr) ||
        (prev_challenge && p->m / 2 < 4))
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(p->ctx->device));
    VMPC_CHECK(p4_lazy_fold_now(p));
    VMPC_CHECK(p4_fold_dots(p, prev_challenge));
    const int rc = p4_round_enqueue(p, nullptr);
    if (rc != VMPC_OK) p->poisoned = true;
    else p->in_flight = true;
    return rc;
}

extern "C" int vmpc_p4_round_end(vmpc_p4 *p, uint8_t out_A[64], uint8_t out_B[64]) {
    if (!p || p->poisoned || !p->in_flight || !out_A || !out_B) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(p->ctx->device));
    p->in_flight = false;
    const int rc = p4_round_collect(p, out_A, out_B);
    if (rc != VMPC_OK) p->poisoned = true;
    return rc;
}

// After the last round's challenge: fold once more and hand back z' (2 x 32 bytes, compressed_pivot.py:77-79).
extern "C" int vmpc_p4_finish(vmpc_p4 *p, const uint8_t last_challenge[32], uint8_t out_z_prime[64]) {
    if (!p || p->poisoned || p->in_flight || !last_challenge || !out_z_prime || p->m != 4 ||
        p->committed != p->total_rounds)
        return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(p->ctx->device));
    VMPC_CHECK(p4_fold_dots(p, last_challenge));
    VMPC_HIP_CHECK(hipMemcpyAsync(out_z_prime, p->z[p->cur], 64, hipMemcpyDeviceToHost, p->ctx->stream));
    return vmpc_ctx_sync(p->ctx);
}

// ---- all rounds behind one call, compact transcript ------------------------------------------------------
// verifiable_mpc_amd/compressed_pivot.py (_Transcript, mode "compact") chains the challenges as
//     state_i = SHA-256(state_{i-1} || round index (4 bytes LE) || A_i (x||y) || B_i (x||y)),   c_i = state_i mod l
// (state as a little-endian integer).  With that chain on this side of the C-ABI the rounds need no interpreter
// between them: the host work per round is the pair's inversion and one 164-byte hash.
namespace {
struct host_sha256 {
    uint32_t h[8];
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t *p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
            0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
            0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
            0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
            0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
            0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
            0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
            0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    // digest of a message of `len` < 2^29 bytes
    static void digest(const uint8_t *msg, size_t len, uint8_t out[32]) {
        host_sha256 s;
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                       0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        memcpy(s.h, iv, 32);
        size_t off = 0;
        for (; off + 64 <= len; off += 64) s.block(msg + off);
        uint8_t tail[128] = {0};
        const size_t rem = len - off;
        memcpy(tail, msg + off, rem);
        tail[rem] = 0x80;
        const size_t blocks = rem < 56 ? 1 : 2;
        const uint64_t bits = (uint64_t)len * 8;
        for (int i = 0; i < 8; i++) tail[64 * blocks - 1 - i] = (uint8_t)(bits >> (8 * i));
        for (size_t b = 0; b < blocks; b++) s.block(tail + 64 * b);
        for (int i = 0; i < 8; i++) {
            out[4 * i] = (uint8_t)(s.h[i] >> 24);
            out[4 * i + 1] = (uint8_t)(s.h[i] >> 16);
            out[4 * i + 2] = (uint8_t)(s.h[i] >> 8);
            out[4 * i + 3] = (uint8_t)s.h[i];
        }
    }
};
}  // namespace

// The chain step: state' = SHA-256(state || round index || A || B), challenge = state' mod l.
static void p4_chain_step(uint8_t state[32], uint32_t round_index, const uint8_t ab[128], uint8_t challenge[32]) {
    uint8_t msg[32 + 4 + 128];
    memcpy(msg, state, 32);
    for (int b = 0; b < 4; b++) msg[32 + b] = (uint8_t)(round_index >> (8 * b));
    memcpy(msg + 36, ab, 128);
    host_sha256::digest(msg, sizeof msg, state);
    uint32_t w[8];
    memcpy(w, state, 32);
    fr_store((uint32_t *)challenge, fr_from_u256(w));
}

// Rounds queued AHEAD of their challenge.  Between two rounds the device used to sit idle for the host's turn:
// hipStreamSynchronize returning (~6 us), the pair's inversion and the hash (~5 us), and the first launch of the
// next round reaching the device (~7 us, the other twenty launches hide behind it) - ~27 us, 19 times per proof.
// Now round i + 1 is queued while round i runs, behind a stream wait (hipStreamWaitValue32) on a word of pinned
// host memory; its two kernels that consume the challenge read it from the pinned block instead of their
// arguments.  The host polls a second pinned word that round i's last kernel sets (vmpc_publish_done), turns A_i, B_i into the challenge, stores it and releases the wait: ~5 us from the store to the
// first kernel (scripts/waitvalue_probe.hip), and the launches of round i + 1 cost the device nothing.
// A round that folds the generators (p4_jump) builds its schedule on the host from the challenges' values and is
// queued the old way, after its challenge; so is everything when the device cannot wait on memory, with a
// communicator (the exchange is a collective on the same stream: kept in the plain order), or on request.
namespace {
struct p4_mailbox {
    volatile uint32_t *done_h, *go_h;
    uint32_t *challenge_h;
    uint32_t *done_d, *go_d;
    const uint32_t *challenge_d;
};
p4_mailbox p4_mail(vmpc_ctx *ctx) {            // behind the results (256 B) and before the status words (2048)
    char *h = (char *)ctx->pin_out, *d = (char *)ctx->pin_out_dev;
    p4_mailbox m;
    m.done_h = (volatile uint32_t *)(h + 1024);
    m.go_h = (volatile uint32_t *)(h + 1088);
    m.challenge_h = (uint32_t *)(h + 1152);
    m.done_d = (uint32_t *)(d + 1024);
    m.go_d = (uint32_t *)(d + 1088);
    m.challenge_d = (const uint32_t *)(d + 1152);
    return m;
}
inline void p4_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// Spin on a pinned completion word.  Every 4096 spins the stream itself is asked: anything but "not ready" /
// "success" is a fault on the queue (a kernel that died never publishes), reported at once instead of after the
// timeout.  1 = the word arrived, 0 = timeout, -1 = stream error.
int p4_poll(volatile uint32_t *word, uint32_t want, double seconds, hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        if (*word == want) return 1;
        p4_cpu_relax();
        if ((spins & 0xfff) == 0xfff) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipSuccess && q != hipErrorNotReady) return -1;
            if (q == hipSuccess && *word != want) {
                // the queue drained and nobody published: a commitment path that did not consume done_flag_dev
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                return *word == want ? 1 : 0;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return 0;
        }
    }
}
}  // namespace

// A queued round is detected by the completion word its commitment's LAST kernel publishes; the launch of that kernel
// takes ctx->done_flag_dev (k_msm_final, k_msm_reduce_combine, k_p4_direct_sum).  A commitment path that does not
// would leave the host polling for 20 s: fail at once instead.
static int p4_flag_consumed(vmpc_ctx *ctx) {
    if (!ctx->done_flag_dev) return VMPC_OK;
    ctx->done_flag_dev = nullptr;
    snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "vmpc_p4_run_compact: the round's commitment did not take the completion word");
    return VMPC_E_HIP;
}

static bool p4_can_queue_ahead(vmpc_p4 *p) {
    if (p->comm || vmpc_getenv_experimental("VMPC_P4_NO_QUEUE_AHEAD")) return false;
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, p->ctx->device) != hipSuccess) return false;
    return can != 0;
}

// state: the chain value before the first round (in), after the last (out).  out_AB: log2(N) - 1 rounds x
// (A_i || B_i) = 128 bytes each; out_z_prime: the two final residues.  The context must be fresh.
extern "C" int vmpc_p4_run_compact(vmpc_p4 *p, uint8_t state[32], int first_round_index, uint8_t *out_AB,
                                   uint8_t out_z_prime[64]) {
    if (!p || !state || !out_AB || !out_z_prime || p->committed != 0 || p->in_flight || p->lazy_fold ||
        first_round_index < 0)
        return VMPC_E_INVAL;             // (lazy folds are asked for by a caller that drives the rounds itself)
    uint8_t challenge[32];
    const int rounds = p->total_rounds;
    if (!p4_can_queue_ahead(p)) {
        for (int i = 0; i < rounds; i++) {
            uint8_t *ab = out_AB + 128 * (size_t)i;
            VMPC_CHECK(vmpc_p4_round(p, i ? challenge : nullptr, ab, ab + 64));
            p4_chain_step(state, (uint32_t)(first_round_index + i), ab, challenge);
        }
        return vmpc_p4_finish(p, challenge, out_z_prime);
    }
    vmpc_ctx *ctx = p->ctx;
    hipStream_t st = ctx->stream;
    // round 0 the plain way: its synchronisation also fetches the status words (canonical z_hat / L~)
    VMPC_CHECK(vmpc_p4_round(p, nullptr, out_AB, out_AB + 64));
    p4_chain_step(state, (uint32_t)first_round_index, out_AB, challenge);
    VMPC_CHECK(vmpc_pinned_reserve(ctx, (size_t)1 << 18));   // nothing in the rounds below stages more than this
    const p4_mailbox mb = p4_mail(ctx);
    bool queued = false;                 // round i sits behind a wait (its fold was queued with a placeholder)
    uint32_t go_seq = 0;
    size_t placeholder_at = 0;
    // a failure with a wait in the queue: let the queue drain (the round runs on a stale challenge; the context is
    // poisoned) before anybody synchronises
    auto bail = [&](int rc) {
        p->poisoned = true;
        if (queued) *mb.go_h = go_seq;
        ctx->stream_waits = false;
        ctx->done_flag_dev = nullptr;
        (void)hipStreamSynchronize(st);
        return rc;
    };
    for (int i = 1; i < rounds; i++) {
        uint8_t *ab = out_AB + 128 * (size_t)i;
        uint32_t done_seq;
        if (queued) {
            // the round is in the queue: hand it its challenge
            memcpy(mb.challenge_h, challenge, 32);
            memcpy(p->pending[placeholder_at].data(), challenge, 32);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            *mb.go_h = go_seq;
            done_seq = go_seq + 1;
            queued = false;
            ctx->stream_waits = false;
        } else {
            int rc = p->m / 2 < 4 ? VMPC_E_INVAL : p4_fold_dots(p, challenge);
            done_seq = ++ctx->p4_seq;
            ctx->done_flag_dev = mb.done_d;                  // the commitment's last kernel publishes the completion
            ctx->done_seq = done_seq;
            if (rc == VMPC_OK) rc = p4_round_enqueue(p, nullptr);
            if (rc == VMPC_OK) rc = p4_flag_consumed(ctx);
            if (rc != VMPC_OK) return bail(rc);
        }
        // queue round i + 1 behind the wait, unless its fold of the witness is followed by a fold of the generators
        const bool jump_next = p->jumps_done < p->jumps.size() &&
                               (int)p->pending.size() + 1 == p->jumps[p->jumps_done].k;
        if (i + 1 < rounds && !jump_next && p->m / 2 >= 4) {
            go_seq = ++ctx->p4_seq;
            ++ctx->p4_seq;                                   // = go_seq + 1: the round's completion word
            int rc = hipStreamWaitValue32(st, mb.go_d, go_seq, hipStreamWaitValueEq, 0xffffffffu) == hipSuccess
                         ? VMPC_OK : VMPC_E_HIP;
            if (rc != VMPC_OK) return bail(rc);
            queued = true;
            ctx->stream_waits = true;
            placeholder_at = p->pending.size();
            rc = p4_fold_dots(p, nullptr, mb.challenge_d);
            ctx->done_flag_dev = mb.done_d;
            ctx->done_seq = go_seq + 1;
            if (rc == VMPC_OK) rc = p4_round_enqueue(p, mb.challenge_d);
            if (rc == VMPC_OK) rc = p4_flag_consumed(ctx);
            if (rc != VMPC_OK) return bail(rc);
        }
        // round i's pair
        const int arrived = p4_poll(mb.done_h, done_seq, 20.0, st);
        if (arrived != 1) {
            snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "vmpc_p4_run_compact: round %d %s", i,
                     arrived < 0 ? "failed on the stream" : "did not publish its completion");
            return bail(VMPC_E_HIP);
        }
        p->committed++;
        p4_affine_pair((const uint8_t *)ctx->pin_out, ab, ab + 64);
        p4_chain_step(state, (uint32_t)(first_round_index + i), ab, challenge);
    }
    return vmpc_p4_finish(p, challenge, out_z_prime);
}
