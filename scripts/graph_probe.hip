// Probe (GPU box): 20 dependent do-nothing kernels per "round", 200 rounds - plain stream launches against one
// hipGraph of the same 20 kernel nodes launched 200 times.  What a graph saves on the DEVICE side per kernel.
// hipcc --offload-arch=gfx950 -O2 scripts/graph_probe.hip -o /tmp/gp && /tmp/gp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_nop(unsigned *p) { if (threadIdx.x == 9999) *p = 1; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned *d;
    CK(hipMalloc(&d, 64));
    const int K = 20, R = 200;
    for (int i = 0; i < 100; i++) k_nop<<<64, 256, 0, st>>>(d);
    CK(hipStreamSynchronize(st));
    double t0 = now_us();
    for (int r = 0; r < R; r++)
        for (int k = 0; k < K; k++) k_nop<<<64, 256, 0, st>>>(d);
    CK(hipStreamSynchronize(st));
    double t_stream = (now_us() - t0) / (R * K);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; k++) k_nop<<<64, 256, 0, st>>>(d);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 10; r++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = now_us();
    for (int r = 0; r < R; r++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    double t_graph = (now_us() - t0) / (R * K);
    printf("per dependent do-nothing kernel: stream launches %.2f us, graph %.2f us\n", t_stream, t_graph);
    return 0;
}
