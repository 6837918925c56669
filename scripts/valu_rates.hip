// Developer microbenchmark (not product): issue rate of the integer instructions the 255-bit field
// arithmetic is made of, chip-wide, so that the "ALU ceiling" of DESIGN.md section 5 can be stated in
// hardware terms (lane-instructions per second against 256 CU x 4 SIMD x 32 lanes x clock).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHAINS 8
#define UNROLL 16

// OP 0: v_mad_u64_u32 (64-bit accumulate)   1: v_mul_lo_u32   2: v_mul_hi_u32   3: v_add_u32
//    4: v_mad_u32_u24                       5: v_fma_f64       6: v_lshl_add_u64 7: v_and_b32
//    8: v_alignbit_b32                      9: v_add_co/v_addc pair (64-bit add)
template <int OP>
__global__ void __launch_bounds__(256) k_rate(const uint32_t *in, uint32_t *out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = in[t & 255] | 1u, b = in[(t + 1) & 255] | 3u;
    uint64_t acc[CHAINS];
    double dacc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        acc[c] = ((uint64_t)in[(t + c) & 255] << 7) | c;
        dacc[c] = (double)c + 0.5;
    }
    const double da = 1.0000001, db = 1e-9;
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (OP == 0) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
                } else if (OP == 1) {
                    uint32_t lo = (uint32_t)acc[c];
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[c] = lo;
                } else if (OP == 2) {
                    uint32_t lo = (uint32_t)acc[c];
                    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[c] = lo;
                } else if (OP == 3) {
                    uint32_t lo = (uint32_t)acc[c];
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[c] = lo;
                } else if (OP == 4) {
                    uint32_t lo = (uint32_t)acc[c];
                    asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo) : "v"(a), "v"(b));
                    acc[c] = lo;
                } else if (OP == 5) {
                    asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(dacc[c]) : "v"(da), "v"(db));
                } else if (OP == 6) {
                    asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(acc[c]) : "v"(acc[(c + 1) % CHAINS]));
                } else if (OP == 7) {
                    uint32_t lo = (uint32_t)acc[c];
                    asm volatile("v_and_b32 %0, %0, %1" : "+v"(lo) : "v"(a));
                    acc[c] = lo;
                } else if (OP == 8) {
                    uint32_t lo = (uint32_t)acc[c];
                    asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo) : "v"(a));
                    acc[c] = lo;
                } else if (OP == 9) {
                    uint32_t lo = (uint32_t)acc[c], hi = (uint32_t)(acc[c] >> 32);
                    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc"
                                 : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
                    acc[c] = ((uint64_t)hi << 32) | lo;
                }
            }
        }
    }
    uint64_t s = 0;
    double ds = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        s += acc[c];
        ds += dacc[c];
    }
    if (s == 0x123456789abcdefull || ds == 1.2345) out[t] = (uint32_t)s;
}

template <int OP>
double run(const char *name, int per_op, const uint32_t *din, uint32_t *dout, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 2000;   // 256-thread blocks = 4 waves: one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_rate<OP><<<blocks, 256>>>(din, dout, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<OP><<<blocks, 256>>>(din, dout, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters * UNROLL * CHAINS * per_op;
    const double rate = n / (ms * 1e-3);
    printf("%-28s %d waves/SIMD: %7.2f T lane-instr/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", name,
           waves_per_simd, rate / 1e12, 256.0 * 4 * 64 * 2.4e9 / rate);
    return rate;
}

int main() {
    uint32_t h[256];
    for (int i = 0; i < 256; i++) h[i] = (uint32_t)(i * 2654435761u + 12345u);
    uint32_t *din, *dout;
    hipMalloc(&din, sizeof h);
    hipMalloc(&dout, (size_t)256 * 8 * 256 * 4);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mad_u64_u32", 1, din, dout, w);
        run<1>("v_mul_lo_u32", 1, din, dout, w);
        run<2>("v_mul_hi_u32", 1, din, dout, w);
        run<3>("v_add_u32", 1, din, dout, w);
        run<4>("v_mad_u32_u24", 1, din, dout, w);
        run<5>("v_fma_f64", 1, din, dout, w);
        run<6>("v_lshl_add_u64", 1, din, dout, w);
        run<7>("v_and_b32", 1, din, dout, w);
        run<8>("v_alignbit_b32", 1, din, dout, w);
        run<9>("v_add_co+v_addc (2 instr)", 2, din, dout, w);
    }
    return 0;
}
