// The one exchange step of the multi-GPU path (SURVEY.md 8e / kernel row "allgather_add_points"): every rank
// contributes k partial commitments (128-byte extended points), an all-gather puts all G x k of them on every
// rank, and every rank adds them IN RANK ORDER with the same kernel (k_points_sum, msm.hip) - bit-identical
// results everywhere, no broadcast.  RCCL has no user-defined reduction, hence gather + ordered add.
//
// Transports behind one vmpc_comm:
//   * RCCL: ncclAllGather on the context's own stream (xGMI inside a node).  librccl is opened with dlopen at the
//     first use - the library has no link-time dependency on it - and an RCCL that is already in the process (torch
//     brings its own, next to the HIP runtime it was built against) is preferred to loading a second one.  The
//     communicator is bootstrapped from a 128-byte unique id that the caller moves between the ranks (any side
//     channel: torch.distributed, MPI, a file).
//   * callback: the caller moves the bytes (host-staged gloo in the tests; any other fabric).  The stream is
//     synchronised before the callback runs; the callback reads `mine` and fills `gathered` (device pointers).
//   * self: world = 1, the gather is a device copy.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>

#include <mutex>

#include "common.h"

namespace {
struct rccl_api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;          // optional: the communicator's own view
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;       // optional
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    int rc = VMPC_E_NODEV;          // outcome of the one load attempt
    char error[256] = {0};          // ... and its diagnostic, repeated to every later caller
};
rccl_api g_rccl;
std::once_flag g_rccl_once;

void rccl_load_once() {
    const char *override_path = getenv("VMPC_RCCL_LIB");
    void *h = nullptr;
    if (override_path && *override_path) {
        h = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
    } else {
        // an RCCL that is already in the process first (torch brings its own, built against the HIP runtime it also
        // brings, under the name "librccl.so"): a second copy from another ROCm release next to it is asking for trouble
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (h) break;
        }
        const char *fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : fresh) {
            if (h) break;
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
    }
    if (!h) {
        const char *why = dlerror();
        snprintf(g_rccl.error, sizeof g_rccl.error, "librccl not found: %s", why ? why : "(no dlerror)");
        return;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(h, "ncclCommUserRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) {
        snprintf(g_rccl.error, sizeof g_rccl.error, "librccl lacks a required symbol");
        dlclose(h);
        return;
    }
    if (getenv("VMPC_DEBUG_STAGES")) {
        Dl_info info;
        if (dladdr((void *)g_rccl.AllGather, &info) && info.dli_fname) fprintf(stderr, "[vmpc] RCCL from %s\n", info.dli_fname);
    }
    g_rccl.handle = h;
    g_rccl.rc = VMPC_OK;
}

// One attempt per process, safe from any number of threads (the in-process tests run one context per thread); a
// failed attempt is remembered WITH its diagnostic, so every later caller gets the same message.
int rccl_load() {
    std::call_once(g_rccl_once, rccl_load_once);
    if (g_rccl.rc != VMPC_OK) snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "%s", g_rccl.error);
    return g_rccl.rc;
}

int rccl_fail(const char *what, ncclResult_t r) {
    snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "%s: %s", what,
             g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return VMPC_E_HIP;
}
}  // namespace

enum { COMM_SELF = 0, COMM_RCCL = 1, COMM_CALLBACK = 2 };

struct vmpc_comm {
    int kind = COMM_SELF;
    int world = 1, rank = 0, device = 0;
    ncclComm_t nccl = nullptr;
    vmpc_exchange_fn fn = nullptr;
    void *user = nullptr;
};

extern "C" int vmpc_comm_unique_id(uint8_t out[VMPC_COMM_ID_BYTES]) {
    if (!out) return VMPC_E_INVAL;
    VMPC_CHECK(rccl_load());
    static_assert(sizeof(ncclUniqueId) == VMPC_COMM_ID_BYTES, "unique id size");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(out, &id, sizeof id);
    return VMPC_OK;
}

extern "C" int vmpc_comm_create_rccl(vmpc_ctx *ctx, const uint8_t unique_id[VMPC_COMM_ID_BYTES], int world, int rank,
                                     vmpc_comm **out) {
    if (!ctx || !unique_id || !out || world < 1 || rank < 0 || rank >= world) return VMPC_E_INVAL;
    VMPC_CHECK(rccl_load());
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    vmpc_comm *c = new vmpc_comm();
    c->kind = COMM_RCCL;
    c->world = world;
    c->rank = rank;
    c->device = ctx->device;
    const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);   // collective: every rank calls it
    if (r != ncclSuccess) {
        delete c;
        return rccl_fail("ncclCommInitRank", r);
    }
    *out = c;
    return VMPC_OK;
}

extern "C" int vmpc_comm_create_callback(int world, int rank, vmpc_exchange_fn fn, void *user, vmpc_comm **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (!fn && world > 1)) return VMPC_E_INVAL;
    vmpc_comm *c = new vmpc_comm();
    c->kind = fn ? COMM_CALLBACK : COMM_SELF;
    c->world = world;
    c->rank = rank;
    c->fn = fn;
    c->user = user;
    *out = c;
    return VMPC_OK;
}

extern "C" int vmpc_comm_destroy(vmpc_comm *c) {
    if (!c) return VMPC_E_INVAL;
    if (c->kind == COMM_RCCL && c->nccl) {
        (void)hipSetDevice(c->device);
        (void)g_rccl.CommDestroy(c->nccl);
    }
    delete c;
    return VMPC_OK;
}

extern "C" int vmpc_comm_info(const vmpc_comm *c, int *world, int *rank, int *kind) {
    if (!c) return VMPC_E_INVAL;
    int w = c->world, r = c->rank;
    if (c->kind == COMM_RCCL && c->nccl) {
        // what RCCL itself says about this communicator, not what the caller passed at creation
        if (g_rccl.CommCount) {
            const ncclResult_t e = g_rccl.CommCount(c->nccl, &w);
            if (e != ncclSuccess) return rccl_fail("ncclCommCount", e);
        }
        if (g_rccl.CommUserRank) {
            const ncclResult_t e = g_rccl.CommUserRank(c->nccl, &r);
            if (e != ncclSuccess) return rccl_fail("ncclCommUserRank", e);
        }
    }
    if (world) *world = w;
    if (rank) *rank = r;
    if (kind) *kind = c->kind;
    return VMPC_OK;
}

extern "C" int vmpc_comm_allgather_dev(vmpc_comm *c, vmpc_ctx *ctx, const void *mine, void *gathered,
                                       size_t bytes_per_rank) {
    if (!c || !ctx || !mine || !gathered || !bytes_per_rank) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "comm_allgather");
    switch (c->kind) {
    case COMM_SELF:
        if (gathered != mine)
            VMPC_HIP_CHECK(hipMemcpyAsync(gathered, mine, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return VMPC_OK;
    case COMM_RCCL: {
        const ncclResult_t r = g_rccl.AllGather(mine, gathered, bytes_per_rank, ncclUint8, c->nccl, ctx->stream);
        return r == ncclSuccess ? VMPC_OK : rccl_fail("ncclAllGather", r);
    }
    case COMM_CALLBACK: {
        VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // host-driven transport: `mine` must be complete
        const int rc = c->fn(c->user, mine, gathered, bytes_per_rank);
        if (rc != 0) {
            snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "exchange callback failed (%d)", rc);
            return VMPC_E_HIP;
        }
        return VMPC_OK;
    }
    }
    return VMPC_E_INVAL;
}

extern "C" int vmpc_comm_points_allsum_dev(vmpc_comm *c, vmpc_ctx *ctx, const void *mine_ext, size_t k,
                                           void *gathered_scratch, void *out_ext, void *out_affine) {
    if (!c || !ctx || !mine_ext || !gathered_scratch || !k || (!out_ext && !out_affine)) return VMPC_E_INVAL;
    VMPC_CHECK(vmpc_comm_allgather_dev(c, ctx, mine_ext, gathered_scratch, VMPC_EXT_BYTES * k));
    // point i of sum j at gathered + 128 (i k + j): exactly the all-gather's rank-major layout
    return vmpc_points_sum_many_dev(ctx, gathered_scratch, (size_t)c->world, k, out_ext, out_affine);
}
