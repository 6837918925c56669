// C-ABI runtime of libvmpc_hip: contexts, memory helpers, profiling, and the host-buffer
// one-shot entry points of include/vmpc.h.
#include <stdlib.h>

#include "common.h"


thread_local char vmpc_err_buf[512] = {0};

// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and
// streams that share a hardware queue run IN ORDER.  This library overlaps work on more streams than that (the main
// stream, the transcript-text stream, the round context's stream, the second commitment of a pair, one stream per
// pipeline slot): with four queues, whether the text stream ended up behind the stream whose folds it is supposed to
// run beside depended on how many contexts had been created before it - 55 instead of 30 ms outside the hash of a
// reference-transcript proof (EXPERIMENTS.md R6.7).  The runtime reads the variable when it initialises, i.e. at the
// process's first HIP call: set here, at load time, unless the host application chose a value (or initialised HIP
// before loading this library - then it has to export the variable itself).
__attribute__((constructor)) static void vmpc_hw_queues_default() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

int vmpc_fr_check_dev(vmpc_ctx *ctx, const void *v, size_t n);  // frvec.hip

extern "C" const char *vmpc_last_error(void) { return vmpc_err_buf; }

const char *vmpc_getenv_experimental(const char *name) {
    const char *on = getenv("VMPC_EXPERIMENTAL");
    return (on && atoi(on) != 0) ? getenv(name) : nullptr;
}

extern "C" int vmpc_backend_info(char *buf, size_t buflen) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (buf && buflen) snprintf(buf, buflen, "no HIP device (%s)", hipGetErrorString(e));
        return VMPC_E_NODEV;
    }
    if (buf && buflen) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, 0) == hipSuccess)
            snprintf(buf, buflen, "%s %s cus=%d lds=%zu hbm=%.0fGiB devices=%d", p.gcnArchName, p.name,
                     p.multiProcessorCount, (size_t)p.sharedMemPerBlock,
                     (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), n);
        else
            snprintf(buf, buflen, "devices=%d", n);
    }
    return n;
}

extern "C" int vmpc_ctx_create(int device, vmpc_ctx **out) {
    if (!out) return VMPC_E_INVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "no HIP device visible");
        return VMPC_E_NODEV;
    }
    if (device < 0 || device >= n) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(device));
    vmpc_ctx *c = new vmpc_ctx();
    c->device = device;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) == hipSuccess) {
        c->cu_count = p.multiProcessorCount;
        c->lds_optin = p.sharedMemPerBlockOptin ? p.sharedMemPerBlockOptin : p.sharedMemPerBlock;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "hipStreamCreate: %s", hipGetErrorString(e));
        return VMPC_E_HIP;
    }
    c->own_stream = true;
    // (+16 words: the arrival counter of vmpc_publish_done sits behind the status words)
    e = hipMalloc((void **)&c->d_status, (VMPC_ST_WORDS + 16) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->d_status, 0, (VMPC_ST_WORDS + 16) * sizeof(uint32_t));
    if (e != hipSuccess) {
        VMPC_IGNORE(hipStreamDestroy(c->stream));
        delete c;
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "hipMalloc(status): %s", hipGetErrorString(e));
        return VMPC_E_HIP;
    }
    const char *w = getenv("VMPC_MSM_WINDOW");
    if (w) c->window_override = atoi(w);
    const char *fj = getenv("VMPC_FOLD_JUMP_DIGITS");    // 4: the fold jump with 4-bit digits (A/B knob)
    if (fj && atoi(fj) == 4) c->fold_jump_digit_bits = 4;
    const char *sp = getenv("VMPC_SHORT_PATH");          // 0: commitments over short 16-row tables take the general path
    if (sp) c->short_path = atoi(sp) != 0;
    // measured and left at their defaults (only with VMPC_EXPERIMENTAL=1):
    const char *bb = vmpc_getenv_experimental("VMPC_BUCKET_BLOCK");
    if (bb && atoi(bb) == 1024) c->bucket_block = 1024;
    const char *bw = vmpc_getenv_experimental("VMPC_BUCKET_WGS_PER_CU");
    if (bw && atoi(bw) >= 0) c->bucket_wgs_per_cu = atoi(bw);
    const char *sf = vmpc_getenv_experimental("VMPC_SORT_FINE_BITS");   // fine bits of the two-level bucket sort
    if (sf && atoi(sf) >= 0) c->sort_fine_bits = atoi(sf);
    const char *rc = vmpc_getenv_experimental("VMPC_REDUCE_MAX_CHUNKS");
    if (rc && atoi(rc) >= 256 && atoi(rc) <= 32768) c->reduce_max_chunks = atoi(rc);
    const char *ro = vmpc_getenv_experimental("VMPC_REDUCE_CHUNKS");
    if (ro && atoi(ro) >= 64 && atoi(ro) <= 32768 && (atoi(ro) & (atoi(ro) - 1)) == 0) c->reduce_chunks_override = atoi(ro);
    const char *rt = vmpc_getenv_experimental("VMPC_REDUCE_TREE");
    if (rt) c->reduce_tree = atoi(rt) != 0;
    const char *ss = vmpc_getenv_experimental("VMPC_SEG_SHIFT_MIN");
    if (ss && atoi(ss) <= 0 && atoi(ss) >= -5) c->seg_shift_min = atoi(ss);
    *out = c;
    return VMPC_OK;
}

extern "C" int vmpc_ctx_destroy(vmpc_ctx *ctx) {
    if (!ctx) return VMPC_E_INVAL;
    if (ctx->p4_pool_busy) {        // a live round context holds this context's arena and a pointer to it
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "vmpc_ctx_destroy: a vmpc_p4 of this context is still alive");
        return VMPC_E_INVAL;
    }
    VMPC_IGNORE(hipSetDevice(ctx->device));
    VMPC_IGNORE(hipStreamSynchronize(ctx->stream));
    for (auto &s : ctx->stages)
        for (auto &pr : s.pending) {
            VMPC_IGNORE(hipEventDestroy(pr.first));
            VMPC_IGNORE(hipEventDestroy(pr.second));
        }
    for (auto e : ctx->event_pool) VMPC_IGNORE(hipEventDestroy(e));
    if (ctx->xevent) VMPC_IGNORE(hipEventDestroy(ctx->xevent));
    if (ctx->ev_sorted) VMPC_IGNORE(hipEventDestroy(ctx->ev_sorted));
    if (ctx->ev_bucketed) VMPC_IGNORE(hipEventDestroy(ctx->ev_bucketed));
    if (ctx->pin_event) VMPC_IGNORE(hipEventDestroy(ctx->pin_event));
    if (ctx->pin) VMPC_IGNORE(hipHostFree(ctx->pin));
    if (ctx->ws) VMPC_IGNORE(hipFree(ctx->ws));
    if (ctx->p4_pool) VMPC_IGNORE(hipFree(ctx->p4_pool));
    if (ctx->p4_kblock) VMPC_IGNORE(hipFree(ctx->p4_kblock));
    if (ctx->d_status) VMPC_IGNORE(hipFree(ctx->d_status));
    if (ctx->short_cursors) VMPC_IGNORE(hipFree(ctx->short_cursors));
    if (ctx->own_stream) VMPC_IGNORE(hipStreamDestroy(ctx->stream));
    delete ctx;
    return VMPC_OK;
}

extern "C" int vmpc_ctx_set_stream(vmpc_ctx *ctx, void *hip_stream) {
    if (!ctx) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) {
        VMPC_IGNORE(hipStreamDestroy(ctx->stream));
        ctx->own_stream = false;
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        VMPC_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return VMPC_OK;
}

extern "C" int vmpc_ctx_sync(vmpc_ctx *ctx) {
    if (!ctx) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    // the status words land in pinned memory (behind the 256 bytes kernels write there): a pageable destination
    // would make this a staged copy, 20 us on every synchronisation
    VMPC_CHECK(vmpc_pinned_reserve(ctx, 0));
    volatile uint32_t *st = (volatile uint32_t *)((char *)ctx->pin_out + 2048);
    VMPC_HIP_CHECK(hipMemcpyAsync((void *)st, ctx->d_status, VMPC_ST_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                  ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (st[VMPC_ST_NONCANON]) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "%u non-canonical scalar(s) (>= l) seen on device",
                 (unsigned)st[VMPC_ST_NONCANON]);
        VMPC_HIP_CHECK(hipMemsetAsync(ctx->d_status, 0, VMPC_ST_WORDS * sizeof(uint32_t), ctx->stream));
        VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return VMPC_E_NONCANON;
    }
    if (st[VMPC_ST_SHORT_OVERFLOW]) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf,
                 "the short-commitment path met scalars beyond its fixed capacities: repeat the call on the general path "
                 "(vmpc_ctx_set_short_path(ctx, 0))");
        VMPC_HIP_CHECK(hipMemsetAsync(ctx->d_status, 0, VMPC_ST_WORDS * sizeof(uint32_t), ctx->stream));
        VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        // scalars like these tend to come again (a witness with many small wires puts every non-zero digit into one
        // bin): the next eligible commitments of this context go straight to the general path
        ctx->short_backoff = 64;
        return VMPC_E_AGAIN;
    }
    return VMPC_OK;
}

// Test hook for the one invariant of the queued-ahead prover (prover.hip): while work sits behind a
// hipStreamWaitValue32 that only this thread can release, nothing may synchronise the stream - so the arena and the
// pinned block must REFUSE to grow (an error, not a deadlock).  Marks the context as holding such a wait without
// queueing one; tests/test_gpu_protocol.py::test_nothing_grows_while_a_wait_is_queued.
extern "C" int vmpc_ctx_debug_hold_wait(vmpc_ctx *ctx, int on) {
    if (!ctx) return VMPC_E_INVAL;
    ctx->stream_waits = on != 0;
    return VMPC_OK;
}

extern "C" int vmpc_ctx_set_short_path(vmpc_ctx *ctx, int on) {
    if (!ctx) return VMPC_E_INVAL;
    ctx->short_path = on ? 1 : 0;
    if (on == 2) ctx->short_backoff = 0;       // 2: on, and forget a recent overflow
    return VMPC_OK;
}

extern "C" int vmpc_ctx_get_short_path(vmpc_ctx *ctx, int *on) {
    if (!ctx || !on) return VMPC_E_INVAL;
    *on = ctx->short_path;
    return VMPC_OK;
}

extern "C" int vmpc_ctx_query(vmpc_ctx *ctx, int *done) {
    if (!ctx || !done) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipSuccess) {
        *done = 1;
    } else if (e == hipErrorNotReady) {
        (void)hipGetLastError();          // not an error: clear the sticky state
        *done = 0;
    } else {
        VMPC_HIP_CHECK(e);
    }
    return VMPC_OK;
}

extern "C" int vmpc_ctx_wait_for(vmpc_ctx *waiter, vmpc_ctx *other) {
    if (!waiter || !other) return VMPC_E_INVAL;
    if (waiter == other || waiter->stream == other->stream) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(other->device));
    if (!other->xevent) VMPC_HIP_CHECK(hipEventCreateWithFlags(&other->xevent, hipEventDisableTiming));
    VMPC_HIP_CHECK(hipEventRecord(other->xevent, other->stream));
    VMPC_HIP_CHECK(hipStreamWaitEvent(waiter->stream, other->xevent, 0));
    return VMPC_OK;
}

// ---- phase pipelining: a bucket stream shared by several contexts ------------------------------------------------
extern "C" int vmpc_stream_create(int device, int priority, void **out) {
    if (!out) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(device));
    int lo = 0, hi = 0;                       // numerically lower = higher priority
    VMPC_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    int pr = priority > 0 ? hi : priority < 0 ? lo : (lo + hi) / 2;
    hipStream_t st = nullptr;
    VMPC_HIP_CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr));
    *out = (void *)st;
    return VMPC_OK;
}

extern "C" int vmpc_stream_destroy(void *hip_stream) {
    if (!hip_stream) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipStreamSynchronize((hipStream_t)hip_stream));
    VMPC_HIP_CHECK(hipStreamDestroy((hipStream_t)hip_stream));
    return VMPC_OK;
}

extern "C" int vmpc_ctx_set_bucket_stream(vmpc_ctx *ctx, void *hip_stream, int wgs_per_cu) {
    if (!ctx || wgs_per_cu < 0 || wgs_per_cu > 8) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    // (no synchronisation: the setting only affects calls made after it, and every call orders its own stages with
    // events - a driver may switch it per call)
    ctx->bucket_stream = (hipStream_t)hip_stream;
    ctx->bucket_wgs_per_cu = wgs_per_cu;
    if (hip_stream && !ctx->ev_sorted) {
        VMPC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_sorted, hipEventDisableTiming));
        VMPC_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_bucketed, hipEventDisableTiming));
    }
    return VMPC_OK;
}

extern "C" int vmpc_ctx_set_window(vmpc_ctx *ctx, int c_bits) {
    if (!ctx || (c_bits != 0 && (c_bits < 4 || c_bits > 16))) return VMPC_E_INVAL;
    ctx->window_override = c_bits;
    return VMPC_OK;
}

int vmpc_pinned_reserve(vmpc_ctx *ctx, size_t bytes) {
    if (ctx->pin && bytes <= ctx->pin_bytes) return VMPC_OK;
    if (ctx->stream_waits) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "pinned block would grow while the stream waits on the host");
        return VMPC_E_INVAL;
    }
    // growing: nothing may still be using the old block (its last H2D is awaited by the caller through pin_event;
    // pin_out is only written by kernels whose results the host has already consumed)
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->pin) VMPC_IGNORE(hipHostFree(ctx->pin));
    ctx->pin = ctx->pin_out = ctx->pin_out_dev = nullptr;
    ctx->pin_bytes = 0;
    const size_t want = ((bytes > 65536 ? bytes : 65536) + 4095) & ~(size_t)4095;
    VMPC_HIP_CHECK(hipHostMalloc(&ctx->pin, want + 4096, hipHostMallocDefault));
    ctx->pin_bytes = want;
    ctx->pin_out = (char *)ctx->pin + want;
    memset(ctx->pin_out, 0, 4096);          // the prover's mailbox words (prover.hip) start below every sequence number
    VMPC_HIP_CHECK(hipHostGetDevicePointer(&ctx->pin_out_dev, ctx->pin_out, 0));
    return VMPC_OK;
}

int vmpc_stage_h2d(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return VMPC_OK;
    if (!ctx->pin_event) VMPC_HIP_CHECK(hipEventCreateWithFlags(&ctx->pin_event, hipEventDisableTiming));
    else VMPC_HIP_CHECK(hipEventSynchronize(ctx->pin_event));     // the previous block has left the buffer
    VMPC_CHECK(vmpc_pinned_reserve(ctx, bytes));
    memcpy(ctx->pin, src, bytes);
    VMPC_HIP_CHECK(hipMemcpyAsync(dst, ctx->pin, bytes, hipMemcpyHostToDevice, ctx->stream));
    VMPC_HIP_CHECK(hipEventRecord(ctx->pin_event, ctx->stream));
    return VMPC_OK;
}

int vmpc_ws_reserve(vmpc_ctx *ctx, size_t total_bytes) {
    ctx->ws_used = 0;
    if (total_bytes <= ctx->ws_bytes) return VMPC_OK;
    if (ctx->stream_waits) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "workspace would grow while the stream waits on the host");
        return VMPC_E_INVAL;
    }
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->ws) {
        VMPC_HIP_CHECK(hipFree(ctx->ws));
        ctx->ws = nullptr;
        ctx->ws_bytes = 0;
    }
    size_t want = total_bytes + total_bytes / 8 + (1 << 20);
    hipError_t e = hipMalloc(&ctx->ws, want);
    if (e != hipSuccess) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "workspace hipMalloc(%zu): %s", want,
                 hipGetErrorString(e));
        return VMPC_E_NOMEM;
    }
    ctx->ws_bytes = want;
    return VMPC_OK;
}

extern "C" int vmpc_malloc(vmpc_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return VMPC_E_NOMEM;
    }
    return VMPC_OK;
}

extern "C" int vmpc_free(vmpc_ctx *ctx, void *dptr) {
    if (!ctx) return VMPC_E_INVAL;
    if (!dptr) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    VMPC_HIP_CHECK(hipFree(dptr));
    return VMPC_OK;
}

extern "C" int vmpc_memcpy_h2d(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return VMPC_E_INVAL;
    if (!bytes) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VMPC_OK;
}

extern "C" int vmpc_memcpy_d2h(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return VMPC_E_INVAL;
    if (!bytes) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VMPC_OK;
}

extern "C" int vmpc_memcpy_d2d(vmpc_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return VMPC_E_INVAL;
    if (!bytes) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return VMPC_OK;
}

// ---- profiling ---------------------------------------------------------------------------------
#define VMPC_PROFILE_EVENTS 2048
extern "C" int vmpc_ctx_profile(vmpc_ctx *ctx, int enable) {
    if (!ctx) return VMPC_E_INVAL;
    if (enable && ctx->event_pool.size() < VMPC_PROFILE_EVENTS) {
        // Create the events NOW, outside whatever the caller is about to time: hipEventCreate is cheap until the
        // runtime has to grow its pool of signals (seen: one 37-ms stall at the ~250th live event of a process,
        // in the middle of a timed region).  2048 events = 1024 stage brackets between two profile_read calls.
        VMPC_HIP_CHECK(hipSetDevice(ctx->device));
        ctx->event_pool.reserve(VMPC_PROFILE_EVENTS);
        while (ctx->event_pool.size() < VMPC_PROFILE_EVENTS) {
            hipEvent_t e = nullptr;
            VMPC_HIP_CHECK(hipEventCreate(&e));
            ctx->event_pool.push_back(e);
        }
    }
    ctx->profile = enable != 0;
    return VMPC_OK;
}

static hipEvent_t take_event(vmpc_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    VMPC_IGNORE(hipEventCreate(&e));
    return e;
}

// VMPC_DEBUG_STAGES=1 (with profiling on): every stage is announced on stderr and synchronised at its end, so a
// kernel that never returns names itself
static bool debug_stages() {
    static const bool on = getenv("VMPC_DEBUG_STAGES") != nullptr;
    return on;
}

int vmpc_stage_begin(vmpc_ctx *ctx, const char *name) {
    if (!ctx->profile) return -1;
    if (debug_stages()) fprintf(stderr, "[vmpc] stage %s ...\n", name);
    int idx = -1;
    for (size_t i = 0; i < ctx->stages.size(); i++)
        if (strcmp(ctx->stages[i].name, name) == 0) idx = (int)i;
    if (idx < 0) {
        vmpc_stage s;
        s.name = name;
        ctx->stages.push_back(s);
        idx = (int)ctx->stages.size() - 1;
    }
    hipEvent_t a = take_event(ctx), b = take_event(ctx);
    VMPC_IGNORE(hipEventRecord(a, ctx->stage_stream ? ctx->stage_stream : ctx->stream));
    ctx->stages[idx].pending.push_back({a, b});
    return idx;
}

void vmpc_stage_end(vmpc_ctx *ctx, int handle) {
    if (handle < 0) return;
    hipStream_t on = ctx->stage_stream ? ctx->stage_stream : ctx->stream;
    VMPC_IGNORE(hipEventRecord(ctx->stages[handle].pending.back().second, on));
    if (debug_stages()) {
        // never synchronise a stream that waits on THIS thread (rounds queued ahead of their challenge,
        // prover.hip): the stage is reported as queued instead
        if (ctx->stream_waits) {
            fprintf(stderr, "[vmpc] stage %s queued behind a challenge wait\n", ctx->stages[handle].name);
            return;
        }
        const hipError_t e = hipStreamSynchronize(on);
        fprintf(stderr, "[vmpc] stage %s done (%s)\n", ctx->stages[handle].name, hipGetErrorString(e));
    }
}

extern "C" int vmpc_ctx_profile_read(vmpc_ctx *ctx, char *names, size_t names_len, double *ms,
                                     uint64_t *launches, int max_stages, int reset) {
    if (!ctx) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->bucket_stream) VMPC_HIP_CHECK(hipStreamSynchronize(ctx->bucket_stream));
    std::string all;
    int k = 0;
    for (auto &s : ctx->stages) {
        for (auto &pr : s.pending) {
            float t = 0;
            if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) {
                s.ms += t;
                s.launches++;
            }
            ctx->event_pool.push_back(pr.first);
            ctx->event_pool.push_back(pr.second);
        }
        s.pending.clear();
        if (k < max_stages) {
            if (ms) ms[k] = s.ms;
            if (launches) launches[k] = s.launches;
            if (k) all += ";";
            all += s.name;
            k++;
        }
        if (reset) {
            s.ms = 0;
            s.launches = 0;
        }
    }
    if (names && names_len) snprintf(names, names_len, "%s", all.c_str());
    return k;
}

// ---- host-buffer one-shots ------------------------------------------------------------------------
namespace {
struct host_call {
    vmpc_ctx *ctx = nullptr;
    std::vector<void *> bufs;
    int rc = VMPC_OK;
    host_call() { rc = vmpc_ctx_create(0, &ctx); }
    ~host_call() {
        if (ctx) {
            for (void *b : bufs) vmpc_free(ctx, b);
            vmpc_ctx_destroy(ctx);
        }
    }
    void *up(const void *src, size_t bytes) {
        if (rc) return nullptr;
        void *d = nullptr;
        rc = vmpc_malloc(ctx, bytes, &d);
        if (rc) return nullptr;
        bufs.push_back(d);
        if (src) rc = vmpc_memcpy_h2d(ctx, d, src, bytes);
        return d;
    }
};
}  // namespace

extern "C" int vmpc_ed25519_msm(const uint8_t *scalars, const uint8_t *points, size_t n,
                                uint8_t out[64]) {
    if (!out || (n && (!scalars || !points))) return VMPC_E_INVAL;
    host_call h;
    void *ds = h.up(scalars, n * 32), *dp = h.up(points, n * 64), *dout = h.up(nullptr, 64);
    if (h.rc) return h.rc;
    uint64_t bad = 0;
    VMPC_CHECK(vmpc_points_validate_dev(h.ctx, dp, n, &bad));
    if (bad) return VMPC_E_NOTONCURVE;
    VMPC_CHECK(vmpc_msm_dev(h.ctx, ds, dp, n, nullptr, nullptr, 0, nullptr, dout));
    VMPC_CHECK(vmpc_ctx_sync(h.ctx));
    return vmpc_memcpy_d2h(h.ctx, out, dout, 64);
}

extern "C" int vmpc_ed25519_fold(const uint8_t *pts_l, const uint8_t *pts_r, const uint8_t c[32],
                                 size_t half, uint8_t *out) {
    if (!c || (half && (!pts_l || !pts_r || !out))) return VMPC_E_INVAL;
    if (!half) return VMPC_OK;
    host_call h;
    void *dl = h.up(pts_l, half * 64), *dr = h.up(pts_r, half * 64), *dout = h.up(nullptr, half * 64);
    if (h.rc) return h.rc;
    uint64_t bad = 0, bad2 = 0;
    VMPC_CHECK(vmpc_points_validate_dev(h.ctx, dl, half, &bad));
    VMPC_CHECK(vmpc_points_validate_dev(h.ctx, dr, half, &bad2));
    if (bad || bad2) return VMPC_E_NOTONCURVE;
    VMPC_CHECK(vmpc_fold_dev(h.ctx, dl, dr, 1, c, half, nullptr, dout));
    VMPC_CHECK(vmpc_ctx_sync(h.ctx));
    return vmpc_memcpy_d2h(h.ctx, out, dout, half * 64);
}

extern "C" int vmpc_ed25519_fixed_base_batch(const uint8_t base[64], const uint8_t *scalars, size_t n,
                                             uint8_t *out) {
    if (!base || (n && (!scalars || !out))) return VMPC_E_INVAL;
    if (!n) return VMPC_OK;
    host_call h;
    void *db = h.up(base, 64), *ds = h.up(scalars, n * 32), *dout = h.up(nullptr, n * 64);
    if (h.rc) return h.rc;
    uint64_t bad = 0;
    VMPC_CHECK(vmpc_points_validate_dev(h.ctx, db, 1, &bad));
    if (bad) return VMPC_E_NOTONCURVE;
    VMPC_CHECK(vmpc_fr_check_dev(h.ctx, ds, n));
    VMPC_CHECK(vmpc_repeat_dev(h.ctx, db, 1, 1, ds, n, 0, nullptr, dout));
    VMPC_CHECK(vmpc_ctx_sync(h.ctx));
    return vmpc_memcpy_d2h(h.ctx, out, dout, n * 64);
}

extern "C" int vmpc_fr_axpy(const uint8_t c[32], const uint8_t *x, const uint8_t *y, size_t n,
                            uint8_t *out) {
    if (!c || (n && (!x || !y || !out))) return VMPC_E_INVAL;
    if (!n) return VMPC_OK;
    host_call h;
    void *dx = h.up(x, n * 32), *dy = h.up(y, n * 32), *dout = h.up(nullptr, n * 32);
    if (h.rc) return h.rc;
    VMPC_CHECK(vmpc_fr_check_dev(h.ctx, dx, n));
    VMPC_CHECK(vmpc_fr_check_dev(h.ctx, dy, n));
    VMPC_CHECK(vmpc_fr_axpy_dev(h.ctx, c, dx, dy, n, dout));
    VMPC_CHECK(vmpc_ctx_sync(h.ctx));
    return vmpc_memcpy_d2h(h.ctx, out, dout, n * 32);
}

extern "C" int vmpc_fr_dot(const uint8_t *a, const uint8_t *b, size_t n, uint8_t out[32]) {
    if (!out || (n && (!a || !b))) return VMPC_E_INVAL;
    host_call h;
    void *da = h.up(a, n * 32), *db = h.up(b, n * 32);
    if (h.rc) return h.rc;
    VMPC_CHECK(vmpc_fr_check_dev(h.ctx, da, n));
    VMPC_CHECK(vmpc_fr_check_dev(h.ctx, db, n));
    VMPC_CHECK(vmpc_fr_dot_dev(h.ctx, da, db, n, out));
    return vmpc_ctx_sync(h.ctx);
}
