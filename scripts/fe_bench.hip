// Developer microbenchmark (not product): field-operation latency and chip-wide throughput on
// gfx950, the yardstick the MSM bucket kernel is priced against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/fe_bench.hip -o /tmp/fe_bench && /tmp/fe_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../verifiable_mpc_amd/csrc/ge25519.h"

template <int V> __device__ __forceinline__ fe mulv(const fe &a, const fe &b) {
    if (V == 0) return fe_mul(a, b);
    return fe_sqr(a);
}

template <int V> __global__ void k_chain(const uint32_t *in, uint32_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe x = fe_load(in + 8 * (i & 1023));
    fe y = fe_load(in + 8 * ((i + 7) & 1023));
    for (int k = 0; k < iters; k++) {
        x = mulv<V>(x, y);
        y = mulv<V>(y, x);
    }
    fe_store(out + 8 * i, fe_add(x, y));
}
// 4 independent chains per thread (ILP as in a point operation)
template <int V> __global__ void k_chain4(const uint32_t *in, uint32_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe a = fe_load(in + 8 * (i & 1023)), b = fe_load(in + 8 * ((i + 1) & 1023));
    fe c = fe_load(in + 8 * ((i + 2) & 1023)), d = fe_load(in + 8 * ((i + 3) & 1023));
    for (int k = 0; k < iters; k++) {
        fe na = mulv<V>(a, b), nb = mulv<V>(b, c), nc = mulv<V>(c, d), nd = mulv<V>(d, a);
        a = na; b = nb; c = nc; d = nd;
    }
    fe_store(out + 8 * i, fe_add(fe_add(a, b), fe_add(c, d)));
}
// mixed additions with a register-resident niels operand: the bucket kernel without memory
__global__ void k_madd(const uint32_t *in, uint32_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    ge_aff a;
    a.x = fe_load(in + 8 * (i & 1023));
    a.y = fe_load(in + 8 * ((i + 5) & 1023));
    ge_niels q = ge_niels_from_affine(a);
    ge_ext p = ge_ext_identity();
    for (int k = 0; k < iters; k++) {
        p = ge_madd(p, q);
        q.t2d.v[0] ^= (uint32_t)k & 1u;
    }
    fe_store(out + 8 * i, fe_add(fe_add(p.X, p.Y), fe_add(p.Z, p.T)));
}

// ---- round 5: the 9-limb alternative --------------------------------------------------------
// 2^255 - 19 in nine limbs of 29 bits (radix 2^29, 261 bits, 2^261 = 1216 mod p): 81 multiply-adds into 17 column
// accumulators instead of 100 into 10 - but the wrap constant no longer fits a pre-multiplied 32-bit operand
// (1216 * 2^29 > 2^32), so the high columns have to be carried down to 29-bit limbs before they can be folded, and
// the result carried once more: two carry passes (17 + 9 columns) and nine more multiply-adds where the 10-limb form
// has one pass of ten.  Correct for reduced inputs (limbs < 2^29 + slack: 9 * 2^58 < 2^64); no room for the lazy sums
// the point formulas of ge25519.h live on (operands up to 2^31).  Product only: a chain of dependent products.
struct fe9 {
    uint32_t v[9];
};
__device__ __forceinline__ fe9 fe9_mul(const fe9 &f, const fe9 &g) {
    uint64_t c[17];
#pragma unroll
    for (int k = 0; k < 17; k++) c[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)f.v[i] * g.v[j];
    // carry the 17 columns down to 29-bit limbs (the last carry is an 18th limb)
    uint32_t r[18];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        const uint64_t t = c[k] + carry;
        r[k] = (uint32_t)t & 0x1fffffffu;
        carry = t >> 29;
    }
    r[17] = (uint32_t)carry;
    // fold limbs 9..17 (weight 2^261 * 2^(29 (k - 9))) with 1216, carry again
    fe9 o;
    carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint64_t t = (uint64_t)r[k] + (uint64_t)r[k + 9] * 1216u + carry;
        o.v[k] = (uint32_t)t & 0x1fffffffu;
        carry = t >> 29;
    }
    o.v[0] += (uint32_t)carry * 1216u;          // < 2^29 + 2^22: within the slack of a reduced limb
    return o;
}
__global__ void k_chain9(const uint32_t *in, uint32_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe9 x, y;
    for (int k = 0; k < 9; k++) {
        x.v[k] = in[8 * (i & 1023) + (k & 7)] & 0x1fffffffu;
        y.v[k] = in[8 * ((i + 7) & 1023) + (k & 7)] & 0x1fffffffu;
    }
    for (int k = 0; k < iters; k++) {
        x = fe9_mul(x, y);
        y = fe9_mul(y, x);
    }
    for (int k = 0; k < 8; k++) out[8 * i + k] = x.v[k] + y.v[k] + (k == 0 ? x.v[8] ^ y.v[8] : 0);
}

template <typename K> double run(K kern, int blocks, int threads, const uint32_t *din, uint32_t *dout, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, threads>>>(din, dout, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, threads>>>(din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    std::vector<uint32_t> h(8 * 1024);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u);
    uint32_t *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, (size_t)8 * 4 * 256 * 1024 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    const int B = 256 * 8, T = 256;
    double lat = run(k_chain<0>, 1, 64, din, dout, iters);
    double thr = run(k_chain<0>, B, T, din, dout, iters);
    double thr4 = run(k_chain4<0>, B, T, din, dout, iters / 2);
    printf("fe_mul  latency %.1f ns (1 wave) | %.1f G/s (chain) | %.1f G/s (4 chains)\n", lat * 1e6 / (2.0 * iters),
           2.0 * iters * B * T / (thr * 1e-3) / 1e9, 4.0 * (iters / 2) * B * T / (thr4 * 1e-3) / 1e9);
    lat = run(k_chain<1>, 1, 64, din, dout, iters);
    thr = run(k_chain<1>, B, T, din, dout, iters);
    thr4 = run(k_chain4<1>, B, T, din, dout, iters / 2);
    printf("fe_sqr  latency %.1f ns (1 wave) | %.1f G/s (chain) | %.1f G/s (4 chains)\n", lat * 1e6 / (2.0 * iters),
           2.0 * iters * B * T / (thr * 1e-3) / 1e9, 4.0 * (iters / 2) * B * T / (thr4 * 1e-3) / 1e9);
    lat = run(k_chain9, 1, 64, din, dout, iters);
    thr = run(k_chain9, B, T, din, dout, iters);
    printf("fe9_mul (9 x 29-bit limbs, product only) latency %.1f ns (1 wave) | %.1f G/s (chain)\n",
           lat * 1e6 / (2.0 * iters), 2.0 * iters * B * T / (thr * 1e-3) / 1e9);
    for (int blocks : {256 * 2, 256 * 3, 256 * 4, 256 * 8}) {
        double m = run(k_madd, blocks, T, din, dout, 500);
        printf("ge_madd %d blocks x 256: %.2f G madd/s  (%.1f G fe_mul-equivalents/s)\n", blocks,
               500.0 * blocks * T / (m * 1e-3) / 1e9, 7 * 500.0 * blocks * T / (m * 1e-3) / 1e9);
    }
    return 0;
}
