// Integer-ALU ceiling of the BN-256 bucket stage (include/vmpc.h vmpc_bn256_madd_rate; bench.py's `alu` block of
// the bn256 line).  A translation unit of its own: instantiating the curve's mixed addition a second time inside
// bn256.hip changed how the compiler built the bucket kernels there (the G2 bucket kernel then never returned).
#include <vector>

#include "common.h"
#include "bn256_curve.h"

#ifndef MSM_BLOCK
#define MSM_BLOCK 256
#endif

// ---- ALU ceiling probe (bench.py `alu` block of the bn256 line): the bucket stage's inner operation - a Jacobian
// mixed addition with an entry in Montgomery form - on registers only, every lane of the chip
template <class C, class F>
__global__ void __launch_bounds__(MSM_BLOCK)
k_bn_madd_rate(const uint32_t *__restrict__ seed, int iters, uint32_t *__restrict__ sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typename C::entry_t e = C::entry_ld(seed);            // arbitrary field elements: the generic branch runs
    typename C::acc_t acc = C::acc_ld(seed + C::ENTRY_WORDS);
    uint32_t w[C::ENTRY_WORDS];
    C::entry_st(w, e);
    w[0] ^= (uint32_t)i & 0xffu;                          // lanes differ
    e = C::entry_ld(w);
    for (int k = 0; k < iters; k++) acc = C::madd(acc, e, (k & 1) != 0);
    uint32_t out[C::ACC_WORDS];
    C::acc_st(out, acc);
    if (out[0] == 0xffffffffu && out[1] == seed[0]) C::acc_st(sink, acc);      // keeps the chain live
}

template <class C, class F>
static int bn_madd_rate(vmpc_ctx *ctx, int iters, double *rate) {
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t words = C::ENTRY_WORDS + 2 * C::ACC_WORDS;
    VMPC_CHECK(vmpc_ws_reserve(ctx, 4 * words + 512));
    uint32_t *buf = (uint32_t *)vmpc_ws_take(ctx, 4 * words);
    std::vector<uint32_t> host(words);
    for (size_t i = 0; i < words; i++) host[i] = 0x01234567u * (uint32_t)(i + 3) & 0x0fffffffu;   // < p in every limb
    VMPC_HIP_CHECK(hipMemcpyAsync(buf, host.data(), 4 * words, hipMemcpyHostToDevice, ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    hipEvent_t e0, e1;
    VMPC_HIP_CHECK(hipEventCreate(&e0));
    VMPC_HIP_CHECK(hipEventCreate(&e1));
    const unsigned blocks = 8u * (unsigned)ctx->cu_count;
    uint32_t *sink = buf + C::ENTRY_WORDS + C::ACC_WORDS;
    k_bn_madd_rate<C, F><<<blocks, MSM_BLOCK, 0, ctx->stream>>>(buf, 4, sink);     // warm-up
    VMPC_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    k_bn_madd_rate<C, F><<<blocks, MSM_BLOCK, 0, ctx->stream>>>(buf, iters, sink);
    VMPC_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    VMPC_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    VMPC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ms <= 0.f) return VMPC_E_HIP;
    *rate = (double)blocks * MSM_BLOCK * (double)iters / (ms * 1e-3);
    return VMPC_OK;
}

extern "C" int vmpc_bn256_madd_rate(vmpc_ctx *ctx, int group, int iters, double *madds_per_second) {
    if (!ctx || !madds_per_second || iters < 1 || (group != 1 && group != 2)) return VMPC_E_INVAL;
    return group == 1 ? bn_madd_rate<G1, BnF1>(ctx, iters, madds_per_second)
                      : bn_madd_rate<G2, BnF2>(ctx, iters, madds_per_second);
}

