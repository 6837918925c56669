// Calibration probe for the HBM-traffic counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports half their bytes on
// gfx950) and says to calibrate any other pattern on a known byte count.  The bucket stage of the MSM (msm.hip
// k_msm_bucket) reads its generators as ONE LANE PER 128-BYTE LINE, eight 16-byte loads, lines at random - this
// kernel does exactly that with a known number of lines, so that the counter's factor for the pattern can be read
// off in the same profiling pass as the bucket kernel itself (scripts/profile_round.sh, scripts/traffic_calibration.py).
#include "common.h"

#define PROBE_BLOCK 256
#define PROBE_PER_LANE 16

__device__ __forceinline__ uint32_t probe_hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// mode 0: lane reads PROBE_PER_LANE random lines; mode 1: consecutive lanes read consecutive lines;
// mode 2: the same bytes as a wide coalesced stream (16 bytes per lane, consecutive lanes consecutive addresses)
__global__ void __launch_bounds__(PROBE_BLOCK, 4)
k_gather_probe(const uint4 *__restrict__ table, size_t table_lines, size_t n_gathers, int mode, uint32_t seed,
               uint4 *__restrict__ sink) {
    const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (mode == 2) {
        const size_t vecs = n_gathers * 8;
        for (size_t v = lane; v < vecs; v += lanes) {
            const uint4 q = table[v % (table_lines * 8)];
            acc.x ^= q.x; acc.y ^= q.y; acc.z ^= q.z; acc.w ^= q.w;
        }
    } else {
        for (size_t g = lane; g < n_gathers; g += lanes) {
            size_t line;
            if (mode == 0) {
                const uint64_t h = ((uint64_t)probe_hash((uint32_t)g ^ seed) << 32) | probe_hash((uint32_t)(g >> 32) + seed * 0x9e3779b9u + (uint32_t)g);
                line = (size_t)(h % table_lines);
            } else {
                line = g % table_lines;
            }
            const uint4 *p = table + line * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint4 q = p[i];
                acc.x ^= q.x; acc.y ^= q.y; acc.z ^= q.z; acc.w ^= q.w;
            }
        }
    }
    // never true for the zero-filled / patterned tables the probe is run on; keeps the loads live
    if (acc.x == 0x9e3779b9u && acc.y == 0x7f4a7c15u && acc.z == seed && acc.w == 0xdeadbeefu) sink[0] = acc;
}

extern "C" int vmpc_gather_probe_dev(vmpc_ctx *ctx, const void *table, size_t table_lines, size_t n_gathers, int mode,
                                     uint32_t seed, double *ms) {
    if (!ctx || !table || !table_lines || !n_gathers || mode < 0 || mode > 2) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    VMPC_CHECK(vmpc_ws_reserve(ctx, 256));
    uint4 *sink = (uint4 *)vmpc_ws_take(ctx, 64);
    size_t blocks = (n_gathers + (size_t)PROBE_BLOCK * PROBE_PER_LANE - 1) / ((size_t)PROBE_BLOCK * PROBE_PER_LANE);
    if (blocks < 1) blocks = 1;
    if (blocks > 0x7fffffffu) return VMPC_E_INVAL;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms) {
        VMPC_HIP_CHECK(hipEventCreate(&e0));
        VMPC_HIP_CHECK(hipEventCreate(&e1));
        VMPC_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    }
    k_gather_probe<<<(unsigned)blocks, PROBE_BLOCK, 0, ctx->stream>>>((const uint4 *)table, table_lines, n_gathers, mode,
                                                                     seed, sink);
    VMPC_KERNEL_CHECK();
    if (ms) {
        VMPC_HIP_CHECK(hipEventRecord(e1, ctx->stream));
        VMPC_HIP_CHECK(hipEventSynchronize(e1));
        float t = 0.f;
        VMPC_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
        *ms = t;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return VMPC_OK;
}
