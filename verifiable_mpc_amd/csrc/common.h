// Shared host-side plumbing of libvmpc_hip: context, workspace arena, error handling,
// per-stage HIP-event profiling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/vmpc.h"

extern thread_local char vmpc_err_buf[512];

#define VMPC_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "%s:%d: %s -> %s", __FILE__, __LINE__, \
                     #expr, hipGetErrorString(_e));                                           \
            return VMPC_E_HIP;                                                                \
        }                                                                                     \
    } while (0)

#define VMPC_CHECK(expr)              \
    do {                              \
        int _r = (expr);              \
        if (_r != VMPC_OK) return _r; \
    } while (0)

// device-side status word layout (ctx->d_status, 4 x u32)
#define VMPC_ST_NONCANON 0  // count of non-canonical scalars seen by recode kernels
#define VMPC_ST_SHORT_OVERFLOW 1  // the fused short-commitment path (msm_short.hip) met a bin or bucket beyond its fixed
                                  // capacities: the call's result is void, vmpc_ctx_sync returns VMPC_E_AGAIN
#define VMPC_ST_WORDS 4

struct vmpc_stage {
    const char *name;
    double ms = 0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct vmpc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // growable scratch arena (never shrinks; reused by every call on this context)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_used = 0;
    uint32_t *d_status = nullptr;
    int window_override = 0;
    int bucket_block = 256;        // threads per workgroup of k_msm_bucket (1024: experiment, msm.hip)
    int bucket_wgs_per_cu = 0;     // > 0: persistent bucket kernel with this many 256-thread workgroups per CU
    int reduce_max_chunks = 32768; // most chunk-lanes per bucket set in the bucket reduction (msm_sort.hip)
    int reduce_chunks_override = 0; // > 0: chunk-lanes per bucket set, fixed (power of two)
    int short_path = 1;            // commitments over a 16-row table of <= 2^17 columns: the fused path (msm_short.hip)
    int short_backoff = 0;         // eligible calls still to take the general path after an overflow (vmpc_ctx_sync sets it)
    size_t lds_optin = 0;          // the device's largest dynamic LDS per workgroup (sharedMemPerBlockOptin)
    int fold_jump_digit_bits = 8;  // signed digits of the fold jump's shared schedule (fold_jump.hip; VMPC_FOLD_JUMP_DIGITS=4: the old form)
    bool sort_wide_ready = false;  // the wide window's fine-sort kernels have their dynamic-LDS limit set (msm_sort.hip)
    bool short_ready = false;      // its kernels' dynamic-LDS limits are set on this context's device
    uint32_t *short_cursors = nullptr;   // [2][128] bin fill counts, zero between calls (the path's last kernel re-arms them)
    int reduce_tree = 1;           // short chunks: weights from the quad tree (msm_reduce_tree.hip), not per-lane ladders
    bool reduce_tree_ready = false; // its kernels' dynamic-LDS limits are set on this context's device
    int seg_shift_min = -3;        // shortest bucket segments the plan may choose: 64 >> 3 entries (msm_sort.hip)
    int sort_fine_bits = -1;       // fine bits of the two-level bucket sort; -1 = automatic (msm_sort.hip)
    int plan_fill_shift = 0;       // the next plan's digit rows are only 1 / 2^shift populated (the A_i, B_i pair of a
                                   // prover round: each scalar vector is zero on half of its positions): the segment
                                   // length follows the expected number of entries, not the number of positions
    int cu_count = 256;
    hipEvent_t xevent = nullptr;   // cross-context ordering (vmpc_ctx_wait_for)
    // Phase pipelining (vmpc_ctx_set_bucket_stream): the bucket stage of every commitment of this context runs on a
    // stream SHARED by the contexts of a pipeline - bucket kernels then run back to back, one at a time, and the sort
    // of the next pass / the reduction and recombination of the previous one run beside them on the contexts' own
    // streams (msm_accumulate).  Not owned.
    hipStream_t bucket_stream = nullptr;
    hipEvent_t ev_sorted = nullptr, ev_bucketed = nullptr;
    hipStream_t stage_stream = nullptr;   // stream the next stage bracket's events are recorded on (nullptr: `stream`)
    // pinned staging for small host -> device parameter blocks (vmpc_stage_h2d)
    void *pin = nullptr;
    size_t pin_bytes = 0;
    hipEvent_t pin_event = nullptr;
    bool stream_waits = false;     // work is queued behind a hipStreamWaitValue32 only this thread can release: nothing
                                   // may synchronise the stream now (vmpc_ws_reserve / vmpc_pinned_reserve refuse to grow)
    // the last kernel of a queued-ahead prover round publishes its completion itself (vmpc_publish_done): set by
    // prover.hip before it queues the round's commitment, consumed (and cleared) by the launch of that kernel
    uint32_t *done_flag_dev = nullptr;
    uint32_t done_seq = 0;
    uint32_t p4_seq = 0;           // sequence numbers of the prover's pinned mailbox words (prover.hip), never reused
    void *pin_out = nullptr;       // 4 KiB of pinned, device-mapped host memory behind `pin` (same allocation): small
    void *pin_out_dev = nullptr;   // results a kernel writes for the host every round (prover.hip); its device address
    // arena of the prover's round context, kept between proofs (prover.hip)
    void *p4_pool = nullptr;
    size_t p4_pool_bytes = 0;
    bool p4_pool_busy = false;
    void *p4_kblock = nullptr;         // the columns of k in a folded vector's table (a single-lane chain of 240
    int p4_kblock_rows = 0;            // doublings, 0.7 ms): k belongs to the CRS, so they are made once
    uint8_t p4_kblock_key[64] = {0};
    // profiling
    bool profile = false;
    std::vector<vmpc_stage> stages;
    std::vector<hipEvent_t> event_pool;
};

// Reserve `bytes` from the context arena (256-byte aligned).  The arena is reset at the
// start of each public call; if it is too small the call grows it (stream-synchronising).
int vmpc_ws_reserve(vmpc_ctx *ctx, size_t total_bytes);
inline void *vmpc_ws_take(vmpc_ctx *ctx, size_t bytes) {
    size_t off = (ctx->ws_used + 255) & ~(size_t)255;
    ctx->ws_used = off + bytes;
    return (char *)ctx->ws + off;
}
// enqueue a copy of a small host block (schedules, parameter tables) to `dst`; `src` may be freed on return
int vmpc_stage_h2d(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes);
// make sure ctx->pin holds `bytes` (>= 64 KiB) and ctx->pin_out / pin_out_dev exist.  ONE allocation for both:
// on this stack a second hipHostMalloc was seen to drop the GPU mapping of an earlier 4-KiB one (memory access
// fault on the first kernel write after it).
int vmpc_pinned_reserve(vmpc_ctx *ctx, size_t bytes);
inline size_t vmpc_align(size_t b) { return (b + 255) & ~(size_t)255; }

// profiling helpers: bracket a kernel launch with events when ctx->profile is on
int vmpc_stage_begin(vmpc_ctx *ctx, const char *name);
void vmpc_stage_end(vmpc_ctx *ctx, int handle);

struct vmpc_stage_scope {
    vmpc_ctx *ctx;
    int h;
    vmpc_stage_scope(vmpc_ctx *c, const char *name) : ctx(c), h(vmpc_stage_begin(c, name)) {}
    ~vmpc_stage_scope() { vmpc_stage_end(ctx, h); }
};

// Tuning knobs whose A/B has been lost (DESIGN.md section 10 keeps the numbers) are read only when
// VMPC_EXPERIMENTAL=1 is set as well: a stray variable in a caller's environment cannot switch a slower path on.
const char *vmpc_getenv_experimental(const char *name);

#define VMPC_IGNORE(expr) ((void)(expr))
#define VMPC_KERNEL_CHECK() VMPC_HIP_CHECK(hipGetLastError())

// Completion word for the host: every workgroup of a grid calls this (one thread, after its own result stores);
// the last one to arrive stores `seq` to `flag` (pinned host memory the host polls) and re-arms the counter.
#ifdef __HIPCC__
__device__ __forceinline__ void vmpc_publish_done(uint32_t *counter, uint32_t *flag, uint32_t seq) {
    __threadfence_system();
    if (atomicAdd(counter, 1u) == gridDim.x - 1) {
        *counter = 0;
        __threadfence_system();
        *(volatile uint32_t *)flag = seq;
    }
}
#endif
