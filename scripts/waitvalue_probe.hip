// Probe (GPU box): host -> device hand-over latency, three ways.
//   A  launch a kernel after the host decides, hipStreamSynchronize for its result        (what vmpc_p4_round does)
//   B  kernel pre-enqueued behind hipStreamWaitValue32 on a pinned flag; host writes the flag, polls a pinned result
//   C  as A but the host polls a pinned result word instead of hipStreamSynchronize
// hipcc --offload-arch=gfx950 -O2 scripts/waitvalue_probe.hip -o /tmp/wv && /tmp/wv
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_answer(const volatile unsigned *in, volatile unsigned *out, unsigned seq) {
    if (threadIdx.x == 0) {
        out[1] = in[1] + 1;
        __threadfence_system();
        out[0] = seq;
    }
}
__global__ void k_busy(unsigned long long cycles, unsigned *sink) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned *hin, *hout, *din, *dout;
    CK(hipHostMalloc(&hin, 64, hipHostMallocMapped));
    CK(hipHostMalloc(&hout, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&din, hin, 0));
    CK(hipHostGetDevicePointer((void **)&dout, hout, 0));
    hin[0] = hin[1] = 0; hout[0] = hout[1] = 0;
    const int N = 200;
    // A
    double sa = 0;
    for (int i = 1; i <= N; i++) {
        k_busy<<<1, 64, 0, st>>>(100 * 100, nullptr);      // ~100 us of "round" (100 MHz wall clock)
        CK(hipStreamSynchronize(st));
        double t0 = now_us();
        hin[1] = i;
        k_answer<<<1, 64, 0, st>>>(din, dout, i);
        CK(hipStreamSynchronize(st));
        sa += now_us() - t0;
        if (hout[1] != (unsigned)i + 1) { printf("A mismatch\n"); return 1; }
    }
    printf("A  launch + hipStreamSynchronize      : %.1f us per hand-over\n", sa / N);
    // C
    double sc = 0;
    for (int i = 1; i <= N; i++) {
        k_busy<<<1, 64, 0, st>>>(100 * 100, nullptr);
        CK(hipStreamSynchronize(st));
        double t0 = now_us();
        hin[1] = 1000 + i;
        k_answer<<<1, 64, 0, st>>>(din, dout, 1000 + i);
        while (*(volatile unsigned *)&hout[0] != 1000u + i) {}
        sc += now_us() - t0;
    }
    CK(hipStreamSynchronize(st));
    printf("C  launch + poll pinned result        : %.1f us per hand-over\n", sc / N);
    if (!can) return 0;
    // B
    double sb = 0;
    for (int i = 1; i <= N; i++) {
        const unsigned seq = 2000 + i;
        k_busy<<<1, 64, 0, st>>>(100 * 100, nullptr);
        CK(hipStreamWaitValue32(st, din, seq, hipStreamWaitValueEq, 0xffffffffu));
        k_answer<<<1, 64, 0, st>>>(din, dout, seq);
        std::this_thread::sleep_for(std::chrono::microseconds(300));   // the "round" is over, the queue sits at the wait
        double t0 = now_us();
        hin[1] = seq;
        __atomic_store_n(&hin[0], seq, __ATOMIC_RELEASE);
        while (*(volatile unsigned *)&hout[0] != seq) {}
        sb += now_us() - t0;
        if (hout[1] != seq + 1) { printf("B mismatch\n"); return 1; }
    }
    CK(hipStreamSynchronize(st));
    printf("B  pre-enqueued behind WaitValue32    : %.1f us per hand-over\n", sb / N);
    // D: as B, completion through hipStreamWriteValue32 behind the kernel (the kernel itself signals nothing)
    double sd = 0;
    for (int i = 1; i <= N; i++) {
        const unsigned seq = 3000 + i;
        k_busy<<<1, 64, 0, st>>>(100 * 100, nullptr);
        CK(hipStreamWaitValue32(st, din, seq, hipStreamWaitValueEq, 0xffffffffu));
        k_answer<<<1, 64, 0, st>>>(din, dout + 4, seq);
        CK(hipStreamWriteValue32(st, dout + 8, seq, 0));
        std::this_thread::sleep_for(std::chrono::microseconds(300));
        double t0 = now_us();
        hin[1] = seq;
        __atomic_store_n(&hin[0], seq, __ATOMIC_RELEASE);
        while (*(volatile unsigned *)&hout[8] != seq) {}
        sd += now_us() - t0;
        if (hout[5] != seq + 1) { printf("D mismatch: result not visible when the written value is\n"); return 1; }
    }
    CK(hipStreamSynchronize(st));
    printf("D  ... completion by WriteValue32      : %.1f us per hand-over\n", sd / N);
    return 0;
}
