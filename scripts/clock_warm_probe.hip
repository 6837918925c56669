// Developer probe (not product): does a latency-bound kernel (one wave, a dependent chain of field products - what a
// small round's exact fold is) run faster when another stream keeps part of the chip busy, i.e. is the 0.78 ms of
// k_fold_quad at 2^2 .. 2^12 elements partly a LOW CLOCK between tiny kernels?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/clock_warm_probe.hip -o /tmp/clock_warm_probe && /tmp/clock_warm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <vector>
#include "../verifiable_mpc_amd/csrc/ge25519.h"

__global__ void k_chain(const uint32_t *in, uint32_t *out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe x = fe_load(in + 8 * (i & 1023)), y = fe_load(in + 8 * ((i + 7) & 1023));
    for (int k = 0; k < iters; k++) {
        x = fe_mul(x, y);
        y = fe_mul(y, x);
    }
    fe_store(out + 8 * i, fe_add(x, y));
}

int main() {
    std::vector<uint32_t> h(8 * 1024);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u);
    uint32_t *din, *dout, *dout2;
    hipMalloc(&din, h.size() * 4);
    hipMalloc(&dout, (size_t)8 * 4 * 256 * 1024);
    hipMalloc(&dout2, (size_t)8 * 4 * 256 * 1024);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto lat = [&](const char *what) {
        float tot = 0;
        const int reps = 10, iters = 1400;               // 2800 dependent products ~ a 253-step ladder of quad operations
        for (int r = 0; r < reps; r++) {
            usleep(1500);                                   // the host side of a small round (hash, glue)
            hipEventRecord(e0, s1);
            k_chain<<<1, 64, 0, s1>>>(din, dout, iters);
            hipEventRecord(e1, s1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            tot += ms;
        }
        printf("%-58s %7.1f us per chain of 2800 products (%5.1f ns each)\n", what, tot / reps * 1e3, tot / reps * 1e6 / 2800);
    };
    k_chain<<<1, 64, 0, s1>>>(din, dout, 10);
    hipDeviceSynchronize();
    lat("chip otherwise idle, 1.5 ms pauses between launches");
    for (int warm_blocks : {8, 32, 128, 1024}) {
        k_chain<<<warm_blocks, 256, 0, s2>>>(din, dout2, 400000);       // busy for a long while on part of the chip
        usleep(20000);
        char buf[96];
        snprintf(buf, sizeof buf, "%d workgroups of field products busy on another stream", warm_blocks);
        lat(buf);
        hipStreamSynchronize(s2);
    }
    return 0;
}
