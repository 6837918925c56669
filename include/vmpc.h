/* vmpc.h - C-ABI of the MI355X-native AC20 hot path (libvmpc_hip.so).
 *
 * The reference (toonsegers/verifiable_mpc) is pure Python and has NO FFI of its own:
 * its seam is the Python functions of verifiable_mpc/ac20/pivot.py and
 * compressed_pivot.py and the MPyC group-element operators they use.
 * Each entry point below names the reference call site whose work it takes over; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and implemented in
 * verifiable_mpc_amd/_native.py.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / Python types;
 *  - scalars: 32 bytes little-endian, canonical residue < l (l = Ed25519 group order);
 *  - affine points: 64 bytes x||y little-endian, canonical residues < p = 2^255-19;
 *  - projective points: 96 bytes X||Y||Z (representative preserved, see ge25519.h);
 *  - extended points: 128 bytes X||Y||Z||T;
 *  - every function returns 0 on success or a negative VMPC_E_* code; the library never
 *    retains host pointers past a call and never draws randomness (the reference draws
 *    r, rho and generator exponents in Python: compressed_pivot.py:105-106,
 *    circuit_sat_r1cs.py:64,81);
 *  - *_dev functions take DEVICE pointers, enqueue on the context's stream and return
 *    without synchronising unless stated; buffers may come from vmpc_malloc or from any
 *    other HIP allocator (e.g. torch tensors' data_ptr()).
 *  - one context per host thread / GPU; no global mutable state besides the HIP runtime.
 */
#ifndef VMPC_H
#define VMPC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMPC_OK 0
#define VMPC_E_INVAL (-22)      /* bad length / null pointer / bad flag */
#define VMPC_E_NONCANON (-34)   /* scalar >= l or coordinate >= p */
#define VMPC_E_NOTONCURVE (-33) /* affine point does not satisfy the curve equation */
#define VMPC_E_NOMEM (-12)
#define VMPC_E_HIP (-5)         /* HIP runtime error, see vmpc_last_error() */
#define VMPC_E_NODEV (-19)      /* no GPU visible */
#define VMPC_E_AGAIN (-11)      /* vmpc_ctx_sync: a commitment took the fused short path (16-row table of <= 2^17 columns)
                                 * and its scalars were skewed beyond that path's fixed capacities (e.g. a witness of
                                 * mostly small values: > ~24 K non-zero digits in one bin); nothing of the call's
                                 * output is valid - switch the path off (vmpc_ctx_set_short_path) and repeat the call.
                                 * The context then skips the short path for its next 64 eligible calls by itself. */

#define VMPC_SCALAR_BYTES 32
#define VMPC_AFFINE_BYTES 64
#define VMPC_PROJ_BYTES 96
#define VMPC_EXT_BYTES 128

typedef struct vmpc_ctx vmpc_ctx;

/* ---- runtime ------------------------------------------------------------------- */
/* number of visible GPUs (or VMPC_E_*); writes "gfx950 MI355X ..." style text */
int vmpc_backend_info(char *buf, size_t buflen);
const char *vmpc_last_error(void);
int vmpc_ctx_create(int device, vmpc_ctx **out);
int vmpc_ctx_destroy(vmpc_ctx *ctx);
/* run on an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own stream */
int vmpc_ctx_set_stream(vmpc_ctx *ctx, void *hip_stream);
int vmpc_ctx_sync(vmpc_ctx *ctx);
/* Commitments over a 16-row fixed-base table of at most 2^17 columns (BASELINE config 2, every prover round after the
 * fold jump) take a fused three-launch path (csrc/msm_short.hip) with fixed capacities; on: the default.  A call whose
 * scalars overflow them is reported by vmpc_ctx_sync as VMPC_E_AGAIN: switch the path off, repeat, switch it on.
 * on = 2: on, and the 64-call back-off that follows an overflow is cleared. */
int vmpc_ctx_set_short_path(vmpc_ctx *ctx, int on);
int vmpc_ctx_get_short_path(vmpc_ctx *ctx, int *on);       /* the current setting (callers that switch it off for a call restore THIS) */
/* test hook: mark the context as holding a queued stream wait (vmpc_p4_run_compact's state between two rounds) -
 * every call that would have to grow the workspace or the pinned block must then fail with VMPC_E_INVAL instead of
 * synchronising a stream that waits on the calling thread */
int vmpc_ctx_debug_hold_wait(vmpc_ctx *ctx, int on);
/* non-blocking: *done = 1 when everything enqueued on the context's stream has completed (a driver that keeps
 * several commitments in flight refills whichever context finishes first) */
int vmpc_ctx_query(vmpc_ctx *ctx, int *done);
/* make `waiter`'s stream wait (on the device, no host block) for everything enqueued so far on
 * `other`'s stream: lets two contexts run independent MSMs (A_i and B_i of one Protocol-4
 * round, compressed_pivot.py:41-42) concurrently */
int vmpc_ctx_wait_for(vmpc_ctx *waiter, vmpc_ctx *other);
int vmpc_malloc(vmpc_ctx *ctx, size_t bytes, void **dptr);
int vmpc_free(vmpc_ctx *ctx, void *dptr);
int vmpc_memcpy_h2d(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes); /* synchronous */
int vmpc_memcpy_d2h(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes); /* synchronous */
int vmpc_memcpy_d2d(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes); /* async */
/* per-stage HIP-event timing of the MSM pipeline (bench.py roofline leg) */
int vmpc_ctx_profile(vmpc_ctx *ctx, int enable);
/* sums since the last reset; names is a ';'-separated list matching ms[] */
int vmpc_ctx_profile_read(vmpc_ctx *ctx, char *names, size_t names_len, double *ms,
                          uint64_t *launches, int max_stages, int reset);
/* Phase pipelining of a stream of commitments (pivot.py:139-145 called proof after proof): the contexts of a pipeline
 * share ONE bucket stream (vmpc_stream_create; priority < 0 = lowest).  Every commitment's bucket stage is then
 * enqueued there - bucket kernels run one at a time, back to back, as persistent launches of `wgs_per_cu` 256-lane
 * workgroups per CU (0 = the whole grid) that leave register-file room on every SIMD - while the sort of the next
 * commitment and the reduction / recombination of the previous one run beside it on the contexts' own streams.
 * Ordering is by events on the device; results and vmpc_ctx_sync behave as before.  NULL = off. */
int vmpc_stream_create(int device, int priority, void **out_hip_stream);
int vmpc_stream_destroy(void *hip_stream);
int vmpc_ctx_set_bucket_stream(vmpc_ctx *ctx, void *hip_stream, int wgs_per_cu);
/* Pippenger window width override (0 = automatic); for tuning / tests */
int vmpc_ctx_set_window(vmpc_ctx *ctx, int c_bits);
/* the window width and window count the planner uses for an n-term Ed25519 MSM */
int vmpc_ed25519_msm_plan(vmpc_ctx *ctx, size_t n, int *c_bits, int *windows);
/* Integer-ALU ceiling of the bucket stage (bench.py roofline leg): mixed additions per second of
 * a register-resident chain on every lane of the chip - the bucket kernel without memory. */
int vmpc_ed25519_madd_rate(vmpc_ctx *ctx, int iters, double *madds_per_second);

/* Calibration probe for the HBM-traffic counters (profiles/, bench.py roofline.traffic): n_gathers reads of one
 * 128-byte line each from a device table of table_lines lines, in the bucket stage's access pattern (one lane per
 * line, eight 16-byte loads).  mode 0: lines at random (hash of the gather index and seed); 1: consecutive lanes
 * read consecutive lines; 2: the same bytes as a wide coalesced stream (16 B per lane).  Known bytes read:
 * 128 * n_gathers.  ms (may be NULL): duration of the launch (synchronises). */
int vmpc_gather_probe_dev(vmpc_ctx *ctx, const void *table, size_t table_lines, size_t n_gathers, int mode,
                          uint32_t seed, double *ms);

/* ---- host-buffer one-shots (SURVEY.md 8b proposal) -------------------------------- */
/* h^gamma-less MSM: out = sum scalars[i] * points[i].
 * Replaces the list comprehension + reduce of pivot.vector_commitment,
 * verifiable_mpc/ac20/pivot.py:143-144 (called from circuit_sat_cb.py:103 and
 * compressed_pivot.py:41,42,110,193). */
int vmpc_ed25519_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[64]);
/* out[i] = c * pts_l[i] + pts_r[i]: the generator fold of
 * verifiable_mpc/ac20/compressed_pivot.py:64 (prover) and :178 (verifier). */
int vmpc_ed25519_fold(const uint8_t *pts_l, const uint8_t *pts_r, const uint8_t c[32],
                      size_t half, uint8_t *out);
/* out[i] = scalars[i] * base: generator setup g_i = h ** r_i,
 * verifiable_mpc/ac20/circuit_sat_r1cs.py:64-70,81. */
int vmpc_ed25519_fixed_base_batch(const uint8_t base[64], const uint8_t *scalars, size_t n,
                                  uint8_t *out);
/* out[i] = c * x[i] + y[i] mod l: z' = z_l + c z_r and L' = c L_l + L_r,
 * verifiable_mpc/ac20/compressed_pivot.py:70-76; z = c0 x + r, :134. */
int vmpc_fr_axpy(const uint8_t c[32], const uint8_t *x, const uint8_t *y, size_t n, uint8_t *out);
/* out = sum a[i] * b[i] mod l: LinearForm evaluation, verifiable_mpc/ac20/pivot.py:84-92. */
int vmpc_fr_dot(const uint8_t *a, const uint8_t *b, size_t n, uint8_t out[32]);

/* ---- device-resident entry points ---------------------------------------------------- */
/* canonical-encoding + on-curve check of n affine points; *n_bad = number of offenders (sync) */
int vmpc_points_validate_dev(vmpc_ctx *ctx, const void *affine, size_t n, uint64_t *n_bad);

/* Pedersen vector commitment as one MSM over n + n_extra terms
 * (pivot.py:139-145: the extra term is h ** gamma).  Any of out_ext / out_affine may be
 * NULL.  Scalars must be canonical (checked on device, reported at the next sync point
 * through vmpc_ctx_sync -> VMPC_E_NONCANON). */
int vmpc_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *affine_points, size_t n,
                 const void *extra_scalars, const void *extra_affine_points, size_t n_extra,
                 void *out_ext, void *out_affine);

/* Fixed-base tables for a generator vector that serves many commitments: the CRS of
 * pivot.py:139-145 (g, h, k are fixed between proofs: circuit_sat_r1cs.py:47-93 creates them once).
 * The table holds rows 2^(256 rho / rows) * P_i, rho = 0..rows-1, rows in {1, 2, 4, 8, 16}, for the n
 * points followed by the n_extra extra points (rows * 128 bytes per point, rows padded to a multiple
 * of 8 points: vmpc_msm_table_bytes).  A commitment over a table needs no point preparation and only
 * (16/rows - 1) * 16 doublings of window recombination (none for rows = 16); its result equals
 * vmpc_msm_dev's on the same points and scalars (as a group element; compared after normalisation).
 * rows = 13 is the WIDE-WINDOW table (round 6): rows 2^(20 rho) * P_i, rho = 0..12, row stride a multiple of 8192
 * points (n + n_extra <= 2^22 - 8192).  A commitment over it is 13 mixed additions per term - the scalar's thirteen signed
 * 20-bit digits, one set of 2^19 buckets - instead of 16, with no recombination at all: the fastest form for the
 * commitments of pivot.py:139-145 from 2^19 generators up (0.98 ms alone at 2^20; 1.66 GB).  Accepted by
 * vmpc_msm_table_dev / _batch_dev and vmpc_p4_set_commit_table; the fold entry points and vmpc_p4_create need one of the
 * other row counts (their kernels read rows spaced 256 / rows bits). */
int vmpc_msm_table_bytes(size_t n, size_t n_extra, int rows, size_t *bytes);
int vmpc_msm_table_build_dev(vmpc_ctx *ctx, const void *affine_points, size_t n,
                             const void *extra_affine_points, size_t n_extra, int rows, void *table);
/* out = sum_{i<m} scalars[i] * P_i + sum_{e<table_extra} extra_scalars[e] * E_e over the table built
 * from (P_0..P_{table_n-1}, E_0..E_{table_extra-1}); m <= table_n; extra_scalars may be NULL (zeros). */
int vmpc_msm_table_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                       const void *scalars, size_t m, const void *extra_scalars, void *out_ext,
                       void *out_affine);
/* `batch` commitments over the same table in one pass (A_i and B_i of a round, compressed_pivot.py:41-42; or
 * independent commitments a prover has queued): scalars[k] / extra_scalars[k] are DEVICE pointers held in HOST
 * arrays of `batch` entries (extra_scalars may be NULL, or hold NULL entries); outputs are consecutive
 * (128 bytes / 64 bytes per commitment).  Same results as `batch` calls of vmpc_msm_table_dev; the bucket
 * reduction and the window recombination - latency chains - run once for the whole batch.  batch <= 16. */
int vmpc_msm_table_batch_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                             const void *const *scalars, size_t m, const void *const *extra_scalars, int batch,
                             void *out_ext, void *out_affine);

/* sum of m extended points in index order, normalised (multi-GPU combine of the per-rank
 * partial commitments; also A * Q^c * B^(c^2) style products once the powers are points) */
int vmpc_points_sum_dev(vmpc_ctx *ctx, const void *ext_points, size_t m, void *out_ext,
                        void *out_affine);
/* k such sums at once: point i of sum j at ext_points + 128 * (i * k + j) - the layout an all-gather of
 * k partial points per rank produces (A_i and B_i of a round: k = 2); outputs consecutive */
int vmpc_points_sum_many_dev(vmpc_ctx *ctx, const void *ext_points, size_t m, size_t k, void *out_ext,
                             void *out_affine);

/* ---- the multi-GPU exchange step (kernel row "allgather_add_points") ------------------------------
 * One process per GPU.  A commitment over generators that are sharded over G ranks (pivot.py:143-144 is a sum of
 * independent terms) is G partial sums plus ONE exchange: an all-gather of the 128-byte partial points and, on
 * every rank, their sum in rank order - the same bits on all ranks, so Fiat-Shamir challenges need no broadcast.
 * Transports: RCCL (ncclAllGather on the context's stream, xGMI inside a node; librccl is dlopen'ed on first use,
 * there is no link-time dependency), or a caller-supplied callback (host-staged gloo in the tests, any other
 * fabric), or none for world = 1. */
typedef struct vmpc_comm vmpc_comm;
#define VMPC_COMM_ID_BYTES 128
/* rank 0 draws the id (ncclGetUniqueId) and hands it to every rank over any side channel */
int vmpc_comm_unique_id(uint8_t out[VMPC_COMM_ID_BYTES]);
/* collective: every rank calls it with the same id (ncclCommInitRank on ctx's device) */
int vmpc_comm_create_rccl(vmpc_ctx *ctx, const uint8_t unique_id[VMPC_COMM_ID_BYTES], int world, int rank,
                          vmpc_comm **out);
/* the caller moves the bytes: fn(user, mine, gathered, bytes_per_rank) is called with DEVICE pointers after the
 * context's stream has been synchronised, must fill gathered[r * bytes_per_rank ..] with rank r's `mine` for every r
 * and return 0.  fn = NULL is allowed for world = 1 (the gather is a device copy). */
typedef int (*vmpc_exchange_fn)(void *user, const void *mine, void *gathered, size_t bytes_per_rank);
int vmpc_comm_create_callback(int world, int rank, vmpc_exchange_fn fn, void *user, vmpc_comm **out);
int vmpc_comm_destroy(vmpc_comm *comm);
/* kind: 0 = self (world 1), 1 = RCCL, 2 = callback.  For an RCCL communicator world / rank are what RCCL itself
 * reports (ncclCommCount / ncclCommUserRank), not the values passed at creation. */
int vmpc_comm_info(const vmpc_comm *comm, int *world, int *rank, int *kind);
/* gathered (world x bytes_per_rank, device) = every rank's `mine`, in rank order; enqueued on ctx's stream */
int vmpc_comm_allgather_dev(vmpc_comm *comm, vmpc_ctx *ctx, const void *mine, void *gathered, size_t bytes_per_rank);
/* k commitments at once: mine_ext = this rank's k partial points (128 B each, device); gathered_scratch = world x k x
 * 128 bytes of device scratch; out_ext / out_affine (either may be NULL) = the k rank-ordered sums, as
 * vmpc_points_sum_many_dev writes them.  No host synchronisation with the RCCL transport. */
int vmpc_comm_points_allsum_dev(vmpc_comm *comm, vmpc_ctx *ctx, const void *mine_ext, size_t k,
                                void *gathered_scratch, void *out_ext, void *out_affine);

/* element-wise `base_i ** n_i` replaying the reference's operation sequence (ge25519.h):
 * bases are projective (96 B) or, with bases_affine != 0, affine (64 B, Z = 1); a single
 * base is broadcast when n_bases == 1.  signed_scalars = 1 applies the reference's
 * pivot._int convention (pivot.py:119-128): residues above l/2 act as negative exponents;
 * signed_scalars = 2 takes exponents in sign-magnitude form (|n| < 2^255, bit 255 = sign) for
 * callers that hold the reference's Python ints, which need not be residues (pivot.py:143).
 * Outputs: projective representatives (may be NULL) and/or affine (may be NULL). */
int vmpc_repeat_dev(vmpc_ctx *ctx, const void *bases, size_t n_bases, int bases_affine,
                    const void *scalars, size_t n, int signed_scalars, void *out_proj,
                    void *out_affine);

/* out_i = scalars[i] * base as AFFINE points (64 B each), for callers that need the group elements of
 * `h ** r_i` (circuit_sat_r1cs.py:64-70,81) but not the reference's projective representatives: a
 * comb table of the one base replaces the per-element ladder (6x less work than vmpc_repeat_dev).
 * Scalars canonical (< l; checked on device, reported at the next sync point). */
int vmpc_fixed_base_dev(vmpc_ctx *ctx, const void *base_affine, const void *scalars, size_t n,
                        void *out_affine);

/* g'_i = (g_l[i] ** c) * g_r[i], compressed_pivot.py:64/:178, replayed exactly.
 * in_affine != 0: inputs are 64-byte affine points (Z = 1), else 96-byte projective. */
int vmpc_fold_dev(vmpc_ctx *ctx, const void *g_l, const void *g_r, int in_affine,
                  const uint8_t c[32], size_t half, void *out_proj, void *out_affine);

/* pivot.list_mul (pivot.py:26-28): balanced pairwise product tree of n projective points
 * in the reference's order, optional identity appended at the end; `points` is clobbered. */
int vmpc_tree_reduce_dev(vmpc_ctx *ctx, void *proj_points, size_t n, int append_identity,
                         void *out_proj);

/* .normalize() for a whole vector (compressed_pivot.py:52,118): projective -> affine */
int vmpc_normalize_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *out_affine);
/* affine (x, y) -> projective (x, y, 1) */
int vmpc_affine_to_proj_dev(vmpc_ctx *ctx, const void *affine, size_t n, void *out_proj);

int vmpc_fr_axpy_dev(vmpc_ctx *ctx, const uint8_t c[32], const void *x, const void *y, size_t n,
                     void *out);
int vmpc_fr_scale_dev(vmpc_ctx *ctx, const uint8_t c[32], const void *x, size_t n, void *out);
/* out[0..n) = c * x + y (y == NULL: c * x), out[n] = tail: z_hat = (c0 * x + r) || phi and L~ = c1 * (L || 0)
 * (compressed_pivot.py:134-141) in one pass each, without a copy to append the scalar.  c, tail: canonical
 * residues (VMPC_E_NONCANON otherwise); out: n + 1 elements. */
int vmpc_fr_axpy_tail_dev(vmpc_ctx *ctx, const uint8_t c[32], const void *x, const void *y, size_t n,
                          const uint8_t tail[32], void *out);
/* out[j] = z[j mod 2^low_bits] * prod_{i < rounds} (c_i if bit (low_bits+rounds-1-i) of j is 0
 * else 1), n = 2^(rounds+low_bits) <= 2^40, rounds <= 20; challenges = rounds x 32 bytes (host).
 * These are the coefficients of `rounds` applications of the fold of compressed_pivot.py:64
 * written as one linear map, so that the verifier's final check (compressed_pivot.py:193) becomes
 * a single N-term MSM over the original generators (used by the compact transcript only). */
int vmpc_fr_challenge_products_dev(vmpc_ctx *ctx, const uint8_t *challenges, int rounds, int low_bits,
                                   const void *z, size_t n, void *out);
/* Scalars of A_i / B_i (compressed_pivot.py:41-42) expressed over a base vector of 2^log2_m0
 * generators to which the last t folds (challenges c_0..c_{t-1}, host, 32 B each) have not been
 * applied; z is the current witness of 2^(log2_m0 - t) scalars.  out_a / out_b: 2^log2_m0 scalars. */
int vmpc_fr_tail_scalars_dev(vmpc_ctx *ctx, const uint8_t *challenges, int t, int log2_m0,
                             const void *z, void *out_a, void *out_b);
/* The same scalars round by round: `products` (2^log2_m0 scalars) carries the challenge products from
 * the call with t - 1 to the call with t (t = 0 initialises it) and only the newest challenge
 * c_{t-1} is passed: two products per element and round instead of up to t + 1. */
int vmpc_fr_tail_scalars_inc_dev(vmpc_ctx *ctx, const uint8_t newest_challenge[32], int t, int log2_m0,
                                 const void *z, void *products, void *out_a, void *out_b);
/* the same for positions j0 .. j0 + count - 1 only; products / out_a / out_b hold `count` elements (one rank's
 * block of g_hat in the sharded prover: scalar work proportional to the block, not to N) */
int vmpc_fr_tail_scalars_block_dev(vmpc_ctx *ctx, const uint8_t newest_challenge[32], int t, int log2_m0,
                                   const void *z, size_t j0, size_t count, void *products, void *out_a,
                                   void *out_b);
/* synchronous: result copied to host */
int vmpc_fr_dot_dev(vmpc_ctx *ctx, const void *a, const void *b, size_t n, uint8_t out[32]);
/* the same, result left in device memory (32 bytes at out_dev, asynchronous) */
int vmpc_fr_dot_to_dev(vmpc_ctx *ctx, const void *a, const void *b, size_t n, void *out_dev);

/* How the reference's str(input_list) prints a curve point is recalled from MPyC, not observed:
 * the bracket pair ('[' ']' or '(' ')') and whether a coordinate c > (p - 1) / 2 prints as -(p - c).  Process-wide,
 * effective for every later formatting call, no rebuild (scripts/check_against_mpyc.py names the call to make when
 * real MPyC prints otherwise).  Scalar signedness travels with each call (is_signed below).  Defaults: '[' ']' 0. */
int vmpc_set_reference_format(char point_open, char point_close, int coord_signed);
int vmpc_get_reference_format(char *point_open, char *point_close, int *coord_signed);
/* Text of the Fiat-Shamir pre-image (pivot.py:134 str(input_list)) produced on device:
 * "item0, item1, ..., item{n-1}, " (every item followed by ", ").  Synchronous; *len gets
 * the number of bytes written, VMPC_E_NOMEM if cap is too small. */
int vmpc_format_points_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *out_text, size_t cap,
                           uint64_t *len);
int vmpc_format_scalars_dev(vmpc_ctx *ctx, const void *scalars, size_t n, int is_signed,
                            void *out_text, size_t cap, uint64_t *len);
/* Asynchronous forms: text is produced into `dev_text` (cap >= n * 245 for points, n * 81 for
 * scalars) and copied, with its length, into PINNED host memory (vmpc_host_alloc) on the context's
 * stream; nothing is valid until that stream has been synchronised (vmpc_ctx_sync). */
int vmpc_format_points_async_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *dev_text, size_t cap,
                                 void *host_text, uint64_t *host_len);
int vmpc_format_scalars_async_dev(vmpc_ctx *ctx, const void *scalars, size_t n, int is_signed,
                                  void *dev_text, size_t cap, void *host_text, uint64_t *host_len);
/* The same with the text delivered in pieces of chunk_bytes (>= 4096): host_len is a pinned block of >= 16 bytes -
 * [0, 8) the text's length, [8, 12) the number of pieces that have landed (0 when the call is made, advanced on the
 * stream) - so that the caller can hash piece k while piece k + 1 is in flight (pivot.py:131-136 hashes ~1 GB of such
 * text per proof at N = 2^20). */
int vmpc_format_points_chunked_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *dev_text, size_t cap,
                                   void *host_text, uint64_t *host_len, size_t chunk_bytes);
int vmpc_format_scalars_chunked_dev(vmpc_ctx *ctx, const void *scalars, size_t n, int is_signed, void *dev_text,
                                    size_t cap, void *host_text, uint64_t *host_len, size_t chunk_bytes);
/* page-locked host memory for the asynchronous copies above */
int vmpc_host_alloc(size_t bytes, void **out);
int vmpc_host_free(void *p);

/* out = A + c * Q + c^2 * B on the HOST, affine x || y (64 bytes) in and out, c a canonical residue: the commitment
 * fold Q' = A * Q**c * B**(c**2) of compressed_pivot.py:66 / :180, whose normalised value the reference transcript
 * hashes every round (no device, no context). */
int vmpc_ed25519_fold_commitment_host(const uint8_t A[64], const uint8_t Q[64], const uint8_t B[64],
                                      const uint8_t c[32], uint8_t out[64]);

/* out = sum_i scalars[i] * points[i] on the HOST (n <= 8; affine 64-byte points, canonical 32-byte residues): the
 * commitment Q = A * P**c0 * k**(c1 (c0 y + t)) of compressed_pivot.py:140 / :233, whose normalised value the first
 * round's pre-image contains (no device, no context). */
int vmpc_ed25519_lincomb_host(const uint8_t *points, const uint8_t *scalars, size_t n, uint8_t out[64]);

/* ---- BN-256 G1 / G2 (Pinocchio prover MSMs) ----------------------------------
 * The eight sums of verifiable_mpc/trinocchio/pynocchio.py:229-246
 *     apply_to_list(point_add, [int(c[i]) * evalkey[...] for i in qap.indices_mid])
 * on the curve of verifiable_mpc/ac20/pairing.py:44-51 (y^2 = x^3 + 3 over F_p; sextic twist over
 * F_p[i]/(i^2+1)).  Scalars: 32 B little-endian, canonical < n (the 256-bit group order).
 * G1 points: 64 B x||y canonical; G2 points: 128 B x.re||x.im||y.re||y.im; the point at infinity
 * is the all-zero encoding.  Outputs are affine. */
int vmpc_bn256_g1_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[64]);
int vmpc_bn256_g2_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[128]);
int vmpc_bn256_g1_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *points, size_t n,
                          void *out_affine);
int vmpc_bn256_g2_msm_dev(vmpc_ctx *ctx, const void *scalars, const void *points, size_t n,
                          void *out_affine);
/* group = 1 (G1) or 2 (G2): canonical encodings and curve equation; *n_bad = offenders (sync) */
/* out_i = scalars[i] * base for one base point (group 1: 64-byte points, group 2: 128-byte): the key generation
 * of the Pinocchio prover is n such products of the two generators (pynocchio.py:101-200).  Scalars: 32-byte
 * little-endian integers below 2^256 (not reduced); affine outputs, all-zero bytes = infinity. */
int vmpc_bn256_fixed_base_dev(vmpc_ctx *ctx, int group, const void *base_affine, const void *scalars, size_t n,
                              void *out_affine);

int vmpc_bn256_validate_dev(vmpc_ctx *ctx, int group, const void *points, size_t n, uint64_t *n_bad);
/* Integer-ALU ceiling of the BN-256 bucket stage (bench.py, `alu` block of the bn256 line): Jacobian mixed
 * additions (madd-2007-bl on the Montgomery-form field; group 2: over F_p^2) per second of a register-resident
 * chain on every lane of the chip - the bucket kernel without memory. */
int vmpc_bn256_madd_rate(vmpc_ctx *ctx, int group, int iters, double *madds_per_second);
/* Fixed-base tables over an evaluation-key vector (fixed per circuit; pynocchio.py:228-246 reads the
 * same evalkey entries for every proof): 17 rows 2^(16 w) * P_i.  vmpc_bn256_table_msm_dev computes
 * sum_{i<m} scalars[i] * P_i, equal to vmpc_bn256_g{1,2}_msm_dev on the first m points. */
int vmpc_bn256_table_bytes(int group, size_t n, size_t *bytes);
int vmpc_bn256_table_build_dev(vmpc_ctx *ctx, int group, const void *points, size_t n, void *table);
/* out_affine (64 / 128 B) and / or out_jacobian (canonical X || Y || Z, 96 / 192 B; Z = 0 is the
 * point at infinity) - the Jacobian form skips the inversion chain that one lane would run */
int vmpc_bn256_table_msm_dev(vmpc_ctx *ctx, int group, const void *table, size_t table_n,
                             const void *scalars, size_t m, void *out_affine, void *out_jacobian);
/* n_tables (<= 16) prepared keys of the SAME length table_n, ONE scalar vector: out_jacobian[k] (96 / 192 B each,
 * consecutive) = sum_{i<m} scalars[i] * P_k[i].  The shape of trinocchio/pynocchio.py:229-246, where the sums
 * r_v*v_mid*g1, r_y*y_mid*g1, r_v*alpha_v*v_mid*g1, ... all run over c_mid: recoding, bucket sort and plan happen once,
 * every table gets its own bucket launch over the one sorted index list, the bucket sets are reduced and finished
 * together.  A column that holds the point at infinity (all-zero entry) in table k contributes nothing to sum k, so
 * sums that differ in a few trailing terms (the zero-knowledge deltas) still share every scalar. */
int vmpc_bn256_table_msm_multi_dev(vmpc_ctx *ctx, int group, const void *const *tables, int n_tables, size_t table_n,
                                   const void *scalars, size_t m, void *out_jacobian);

/* SHA-256 of every `chunk_bytes`-sized piece of a device buffer (last piece may be short):
 * out_digests[i] = SHA256(data[i*chunk : (i+1)*chunk]), 32 bytes each.  Leaves of the compact
 * transcript's two-level digests; not used by the reference transcript. */
int vmpc_sha256_chunks_dev(vmpc_ctx *ctx, const void *data, size_t nbytes, size_t chunk_bytes,
                           void *out_digests);

/* k halving folds of a tabulated generator vector in one pass (compressed_pivot.py:64 applied k times):
 *   out[j] = sum_{b < 2^k} scalars[b] * P[j + b * (n_cols >> k)],   j < n_cols >> k,
 * over the first n_cols columns of `table` (generators, then extras), n_cols a power of two, 1 <= k <= 6.
 * `scalars`: 2^k canonical 32-byte residues in HOST memory (the challenge products); out: affine x||y.
 * 64 mixed additions per column instead of a 253-bit scalar multiplication per generator and round. */
int vmpc_msm_table_fold_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                            size_t n_cols, int k, const uint8_t *scalars, void *out_affine);

/* the same fold, leaving the folded vector's own fixed-base table (out_rows rows, the n_extra device points
 * extra_affine_points as its extras; size: vmpc_msm_table_bytes(n_cols >> k, n_extra, out_rows)) instead of the
 * vector - what a prover that keeps committing to the folded generators wants (vmpc_p4_round uses it) */
int vmpc_msm_table_fold_table_dev(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows,
                                  size_t n_cols, int k, const uint8_t *scalars, const void *extra_affine_points,
                                  size_t n_extra, int out_rows, void *out_table);

/* ---- device-resident Protocol-4 prover rounds (SURVEY.md 8b "vmpc_ctx_round") -----------------------------------
 * One halving round of compressed_pivot.py:29-86 per call: z_hat, L~ and the per-generator challenge products stay in
 * HBM for all log N rounds; a round folds z_hat and L~ with the previous challenge (:70-76), computes the two
 * inner products that are the exponents of k (:41-42), and A_i, B_i as one batched pass over the tabulated CRS
 * (the generators are not folded: the pending challenge products multiply the scalars).  The Fiat-Shamir hash
 * stays with the caller.  `table`: vmpc_msm_table_build_dev over table_n generators g followed by table_extra
 * extras; extras 0 .. h_slots-1 are the tail of g_hat (h), extra k_slot is k (k_affine: the same point, x||y,
 * host memory).  z_hat / L_tilde: N = table_n + h_slots device scalars each, N a power of two.
 * On a CRS of 2^18 generators or more the context folds the generators once, after 5 rounds, with
 * vmpc_msm_table_fold_dev and continues on a table of the 32-times shorter vector (VMPC_P4_JUMP=0: never). */
typedef struct vmpc_p4 vmpc_p4;
int vmpc_p4_create(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows, int h_slots,
                   int k_slot, const uint8_t k_affine[64], const void *z_hat, const void *L_tilde, vmpc_p4 **out);
/* vmpc_p4_create with the number of rounds before the generators are folded chosen by the caller (jump_k = 0: never,
 * < 0: the default of 5) and, with lazy_fold != 0, a fold that waits for vmpc_p4_prefold: the round that is given the
 * jump_k-th challenge still commits over the unfolded table, and the fold is enqueued - on the context's stream,
 * nothing waited for - when the caller asks (not between vmpc_p4_round_begin and _end), at the latest at the start of
 * the round after.  vmpc_p4_prefold without a fold that is due does nothing.  vmpc_p4_run_compact refuses a lazy
 * context. */
int vmpc_p4_create_opts(vmpc_ctx *ctx, const void *table, size_t table_n, size_t table_extra, int rows, int h_slots,
                        int k_slot, const uint8_t k_affine[64], const void *z_hat, const void *L_tilde, int jump_k,
                        int lazy_fold, vmpc_p4 **out);
int vmpc_p4_prefold(vmpc_p4 *p4);
/* A second table over the SAME generators and extras (same n, same extras in the same order) for the A_i, B_i
 * commitments of the rounds BEFORE the generators are folded - the 13-row wide-window table (vmpc_msm_table_build_dev
 * with rows = 13): 13 mixed additions per term instead of 16.  The fold itself reads the table given to
 * vmpc_p4_create (rows spaced 256 / rows bits).  NULL: none.  The caller keeps it alive as long as the context. */
int vmpc_p4_set_commit_table(vmpc_p4 *p4, const void *table, int rows);
/* The same rounds with g_hat cut into `world` CONTIGUOUS blocks, one per rank of `comm` (one process
 * per GPU): block_table = vmpc_msm_table_build_dev over THIS rank's block_n = N / world generators (h is the last
 * generator of the last block) with k among its extras (k_slot); z_hat / L_tilde: all N scalars, the same on every
 * rank.  A round computes the partial A_i, B_i over the block and exchanges them once (vmpc_comm_points_allsum_dev);
 * every rank returns the same A_i, B_i.  The k term is added by rank 0.  The fold of the generators after 5 rounds
 * stays local: a rank folds the strides its block holds (needs 2^5 >= 2 world and a block of >= 2^18 generators)
 * and later rounds commit to the rank's partial vector.  world must be a power of two.  All ranks must make the
 * same calls in the same order. */
int vmpc_p4_create_sharded(vmpc_ctx *ctx, vmpc_comm *comm, const void *block_table, size_t block_n, size_t table_extra,
                           int rows, int k_slot, const uint8_t k_affine[64], const void *z_hat, const void *L_tilde,
                           vmpc_p4 **out);
/* prev_challenge: derived from the previous call's A, B; NULL on the first call.  out_A / out_B: affine x||y.
 * Valid log2(N) - 1 times.  If a call fails after the witness fold was enqueued the context is unusable: every
 * further call returns VMPC_E_INVAL, destroy it (vmpc_ctx_destroy refuses while a round context is alive). */
int vmpc_p4_round(vmpc_p4 *p4, const uint8_t prev_challenge[32], uint8_t out_A[64], uint8_t out_B[64]);
/* vmpc_p4_round in two halves, for a caller with host work of its own to do while the pair is computed (the reference
 * transcript's prover enqueues the round's exact generator fold on another stream meanwhile): _begin folds the witness
 * with prev_challenge and enqueues the commitments without waiting, _end waits and returns A_i, B_i.  Until _end every
 * other vmpc_p4_* call on this context except vmpc_p4_destroy returns VMPC_E_INVAL. */
int vmpc_p4_round_begin(vmpc_p4 *p4, const uint8_t prev_challenge[32]);
int vmpc_p4_round_end(vmpc_p4 *p4, uint8_t out_A[64], uint8_t out_B[64]);
/* after the last round: fold with its challenge and return z' (two 32-byte residues, compressed_pivot.py:77-79) */
int vmpc_p4_finish(vmpc_p4 *p4, const uint8_t last_challenge[32], uint8_t out_z_prime[64]);
/* every round and the finish behind one call, with the COMPACT transcript's challenge chain (verifiable_mpc_amd/
 * compressed_pivot.py, _Transcript): state_i = SHA-256(state_{i-1} || round index, 4 bytes LE || A_i || B_i),
 * c_i = state_i (little-endian integer) mod l.  state: in = the chain value before the first round, out = after the
 * last.  out_AB: (log2(N) - 1) x 128 bytes, A_i || B_i affine.  The context must not have run a round yet. */
int vmpc_p4_run_compact(vmpc_p4 *p4, uint8_t state[32], int first_round_index, uint8_t *out_AB, uint8_t out_z_prime[64]);
int vmpc_p4_destroy(vmpc_p4 *p4);

#ifdef __cplusplus
}
#endif
#endif
