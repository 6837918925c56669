// Developer microbenchmark (not product), VERDICT r05 item 5: is batched-affine bucket accumulation worth building for
// the BN-256 sums of the Pinocchio prover (verifiable_mpc/trinocchio/pynocchio.py:229-246)?  Register / LDS resident
// rate of each form over sw256.h's Montgomery field, no memory traffic:
//   J  Jacobian mixed addition madd-2007-bl (7M + 4S), what gk_bucket does today: one dependent chain per lane
//   A  affine additions with ONE inversion per workgroup and batch (Montgomery's trick): every lane takes B independent
//      pairs, keeps the running products of their x-differences, the 256 lane totals go up a product tree in LDS, one
//      lane inverts the root (square-and-multiply, 384 products), the inverses come down the tree and every lane
//      back-substitutes: 1 + 2 + 3 = 6 products per addition + the tree's 3 per lane and batch
// (exceptional cases - equal x - are not handled: timing only)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/bn_affine_bench.hip -o gpurun_out/bn_affine_bench && gpurun_out/bn_affine_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../verifiable_mpc_amd/csrc/sw256.h"

#define WG 256

__device__ __forceinline__ fp ld(const uint32_t *p) {
    fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = p[i];
    return r;
}
__device__ __forceinline__ void st(uint32_t *p, const fp &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = a.v[i];
}

__global__ void __launch_bounds__(WG) k_jac(const uint32_t *in, uint32_t *out, int iters) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    aff<Fp1Ops> q;
    q.x = ld(in + 8 * (i & 255));
    q.y = ld(in + 8 * ((i + 5) & 255));
    q.inf = false;
    jac<Fp1Ops> acc;
    acc.X = ld(in + 8 * ((i + 9) & 255));
    acc.Y = ld(in + 8 * ((i + 17) & 255));
    acc.Z = ld(in + 8 * ((i + 33) & 255));
    for (int k = 0; k < iters; k++) {
        acc = jac_madd<Fp1Ops>(acc, q);
        q.x.v[0] ^= (uint32_t)k & 1u;
    }
    st(out + 8 * i, fp_add(fp_add(acc.X, acc.Y), acc.Z));
}

// B additions per lane and batch; P[] = running products of the x-differences (registers for B <= 8, else LDS)
template <int B>
__global__ void __launch_bounds__(WG) k_affine(const uint32_t *in, uint32_t *out, int batches) {
    extern __shared__ uint32_t lds[];
    uint32_t *tree = lds;                                        // [2 * WG] field elements: product tree
    uint32_t *pp = lds + 2 * WG * 8;                             // [WG][B] running products when they do not fit registers
    const int t = threadIdx.x;
    const size_t gi = (size_t)blockIdx.x * blockDim.x + t;
    fp x1 = ld(in + 8 * (gi & 255)), y1 = ld(in + 8 * ((gi + 5) & 255));
    fp x2 = ld(in + 8 * ((gi + 9) & 255)), y2 = ld(in + 8 * ((gi + 17) & 255));
    fp sink = fp_zero();
    for (int it = 0; it < batches; it++) {
        fp P[B <= 8 ? B : 1];
        fp run = fp_one();
        // pair j of this lane: (x1 + j, y1) + (x2, y2 + j)   (cheap, distinct inputs)
#pragma unroll
        for (int j = 0; j < B; j++) {
            fp a = x1;
            a.v[0] += (uint32_t)j;
            const fp d = fp_sub(x2, a);
            run = fp_mul(run, d);
            if (B <= 8) P[j] = run;
            else st(pp + ((size_t)t * B + j) * 8, run);
        }
        // product tree over the workgroup's 256 totals: leaves at tree[WG + t]
        st(tree + (WG + t) * 8, run);
        __syncthreads();
        for (int w = WG / 2; w >= 1; w >>= 1) {
            if (t < w) st(tree + (w + t) * 8, fp_mul(ld(tree + (2 * (w + t)) * 8), ld(tree + (2 * (w + t) + 1) * 8)));
            __syncthreads();
        }
        if (t == 0) st(tree + 8, fp_inv(ld(tree + 8)));          // the ONE inversion
        __syncthreads();
        for (int w = 1; w < WG; w <<= 1) {                        // node n holds its subtree's product; inverse comes down
            fp l, r, inv;
            if (t < w) {
                inv = ld(tree + (w + t) * 8);
                l = ld(tree + (2 * (w + t)) * 8);
                r = ld(tree + (2 * (w + t) + 1) * 8);
            }
            __syncthreads();
            if (t < w) {
                st(tree + (2 * (w + t)) * 8, fp_mul(inv, r));
                st(tree + (2 * (w + t) + 1) * 8, fp_mul(inv, l));
            }
            __syncthreads();
        }
        fp inv = ld(tree + (WG + t) * 8);                        // 1 / (this lane's total)
#pragma unroll
        for (int j = B - 1; j >= 0; j--) {
            fp a = x1;
            a.v[0] += (uint32_t)j;
            const fp d = fp_sub(x2, a);
            const fp prev = j ? (B <= 8 ? P[j ? j - 1 : 0] : ld(pp + ((size_t)t * B + j - 1) * 8)) : fp_one();
            const fp dinv = fp_mul(inv, prev);
            inv = fp_mul(inv, d);
            fp b = y2;
            b.v[0] += (uint32_t)j;
            const fp lam = fp_mul(fp_sub(b, y1), dinv);
            const fp x3 = fp_sub(fp_sub(fp_sqr(lam), a), x2);
            const fp y3 = fp_sub(fp_mul(lam, fp_sub(a, x3)), y1);
            sink = fp_add(sink, fp_add(x3, y3));
        }
        x1 = fp_add(x1, sink);                                   // next batch's inputs depend on this one's results
        __syncthreads();
    }
    st(out + 8 * gi, sink);
}

template <typename F> static double timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    std::vector<uint32_t> h(8 * 256);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u) & ((i % 8 == 7) ? 0x3fffffffu : 0xffffffffu);
    uint32_t *din, *dout;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, (size_t)8 * 4 * 256 * 8192);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    {
        const int blocks = 8 * cus, iters = 400;
        double ms = timed([&](int warm) { k_jac<<<blocks, WG>>>(din, dout, warm ? 4 : iters); });
        printf("Jacobian madd-2007-bl (7M + 4S), dependent chain per lane: %7.2f G additions/s\n",
               (double)blocks * WG * iters / (ms * 1e-3) / 1e9);
    }
    auto run_affine = [&](auto kern, int B, int wgs_per_cu) {
        const size_t lds = (size_t)(2 * WG * 8 + (B > 8 ? WG * B * 8 : 0)) * 4;
        if (lds > (size_t)pr.sharedMemPerBlockOptin && lds > (size_t)pr.sharedMemPerBlock) {
            printf("affine, B = %2d: %zu KB of LDS per workgroup do not fit (skipped)\n", B, lds >> 10);
            return;
        }
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int blocks = wgs_per_cu * cus, batches = 24;
        double ms = timed([&](int warm) { kern<<<blocks, WG, lds>>>(din, dout, warm ? 1 : batches); });
        printf("affine, one inversion per workgroup and batch, B = %2d pairs per lane, %2d workgroups per CU: %7.2f G additions/s"
               "  (LDS %zu KB per workgroup)\n", B, wgs_per_cu, (double)blocks * WG * B * batches / (ms * 1e-3) / 1e9, lds >> 10);
    };
    for (int w : {4, 8, 16}) run_affine(k_affine<8>, 8, w);
    for (int w : {2, 4}) run_affine(k_affine<16>, 16, w);
    for (int w : {1, 2}) run_affine(k_affine<32>, 32, w);
    return 0;
}
