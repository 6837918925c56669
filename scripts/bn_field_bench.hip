// Developer microbenchmark (not product): dependent-chain latency and chip-wide throughput of the two BN-256 base
// fields - sw256.h's saturated product scan (throughput kernels) and fp29.h's 9 x 29-bit limbs (latency kernels).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/bn_field_bench.hip -o gpurun_out/bn_field_bench && gpurun_out/bn_field_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../verifiable_mpc_amd/csrc/fp29.h"

template <int V> __global__ void k_chain(const uint32_t *in, uint32_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (V == 0) {
        fp x = Fp1Ops::load_raw(in + 8 * (i & 1023)), y = Fp1Ops::load_raw(in + 8 * ((i + 7) & 1023));
        for (int k = 0; k < iters; k++) {
            x = fp_mul(x, y);
            y = fp_mul(y, x);
        }
        Fp1Ops::store_raw(out + 8 * i, fp_add(x, y));
    } else if (V == 1) {
        fp29 x = Fp29Ops::load_raw(in + 8 * (i & 1023)), y = Fp29Ops::load_raw(in + 8 * ((i + 7) & 1023));
        for (int k = 0; k < iters; k++) {
            x = fp29_mul(x, y);
            y = fp29_mul(y, x);
        }
        Fp29Ops::store_raw(out + 8 * i, fp29_add(x, y));
    } else if (V == 2) {        // additions / subtractions in the chain, as in a point formula
        fp x = Fp1Ops::load_raw(in + 8 * (i & 1023)), y = Fp1Ops::load_raw(in + 8 * ((i + 7) & 1023));
        for (int k = 0; k < iters; k++) {
            x = fp_mul(fp_add(x, y), fp_sub(x, y));
            y = fp_mul(fp_sub(y, x), fp_add(y, y));
        }
        Fp1Ops::store_raw(out + 8 * i, fp_add(x, y));
    } else {
        fp29 x = Fp29Ops::load_raw(in + 8 * (i & 1023)), y = Fp29Ops::load_raw(in + 8 * ((i + 7) & 1023));
        for (int k = 0; k < iters; k++) {
            x = fp29_mul(fp29_add(x, y), fp29_sub(x, y));
            y = fp29_mul(fp29_sub(y, x), fp29_add(y, y));
        }
        Fp29Ops::store_raw(out + 8 * i, fp29_add(x, y));
    }
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
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u) & ((i % 8 == 7) ? 0x3fffffffu : 0xffffffffu);
    uint32_t *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, (size_t)8 * 4 * 256 * 1024 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 1000, B = 256 * 8, T = 256;
    const char *names[4] = {"fp_mul   (8 x 32, product scan)", "fp29_mul (9 x 29, columns)     ", "fp   mul + add + sub           ", "fp29 mul + add + sub           "};
    double (*runs[4])(int, int, const uint32_t *, uint32_t *, int) = {
        [](int b, int t, const uint32_t *i, uint32_t *o, int n) { return run(k_chain<0>, b, t, i, o, n); },
        [](int b, int t, const uint32_t *i, uint32_t *o, int n) { return run(k_chain<1>, b, t, i, o, n); },
        [](int b, int t, const uint32_t *i, uint32_t *o, int n) { return run(k_chain<2>, b, t, i, o, n); },
        [](int b, int t, const uint32_t *i, uint32_t *o, int n) { return run(k_chain<3>, b, t, i, o, n); }};
    for (int v = 0; v < 4; v++) {
        double lat = runs[v](1, 64, din, dout, iters);
        double lat4 = runs[v](256, 256, din, dout, iters);      // one wave per SIMD
        double thr = runs[v](B, T, din, dout, iters);
        printf("%s  %.0f ns per dependent step (1 wave alone) | %.0f ns (1 wave per SIMD) | %.1f G steps/s chip-wide\n", names[v],
               lat * 1e6 / (2.0 * iters), lat4 * 1e6 / (2.0 * iters), 2.0 * iters * B * T / (thr * 1e-3) / 1e9);
    }
    return 0;
}
