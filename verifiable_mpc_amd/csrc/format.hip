// Device-side production of the Fiat-Shamir pre-image text.
//
// The reference hashes str(input_list) (verifiable_mpc/ac20/pivot.py:131-136), which for
// Protocol 4 contains every generator of the round and every coefficient of the linear
// form (compressed_pivot.py:51-59) as decimal integers: ~250 MB of text at N = 2^20,
// round 0.  Two kernels per vector: item lengths -> exclusive scan -> write at offsets.
// Output layout: "item0, item1, ..., item{n-1}, " (each item followed by ", "); the host
// feeds SHA-256 with the brackets and without the last separator.
#include "common.h"
#include "fmt.h"
#include "scan.h"

#define FMT_BLOCK 256

// The [mpyc-recall] format choices of the PRODUCT, process-wide (vmpc_set_reference_format): how the reference's
// str(input_list) prints a curve point.  Scalar signedness travels with every call (is_signed).
static fmt_point_style g_point_style = {'[', ']', 0};

extern "C" int vmpc_set_reference_format(char point_open, char point_close, int coord_signed) {
    const bool ok = (point_open == '[' && point_close == ']') || (point_open == '(' && point_close == ')');
    if (!ok || (coord_signed != 0 && coord_signed != 1)) return VMPC_E_INVAL;
    g_point_style.open = point_open;
    g_point_style.close = point_close;
    g_point_style.coord_signed = coord_signed;
    return VMPC_OK;
}

extern "C" int vmpc_get_reference_format(char *point_open, char *point_close, int *coord_signed) {
    if (point_open) *point_open = g_point_style.open;
    if (point_close) *point_close = g_point_style.close;
    if (coord_signed) *coord_signed = g_point_style.coord_signed;
    return VMPC_OK;
}

__device__ __forceinline__ void fmt_ld8(uint32_t d[8], const uint32_t *src) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    uint4 a = p[0], b = p[1];
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
    d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}

__global__ void __launch_bounds__(FMT_BLOCK)
k_fmt_points_len(const uint32_t *__restrict__ proj, size_t n, fmt_point_style style, uint32_t *__restrict__ lens) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t X[8], Y[8], Z[8];
    fmt_ld8(X, proj + 24 * i);
    fmt_ld8(Y, proj + 24 * i + 8);
    fmt_ld8(Z, proj + 24 * i + 16);
    lens[i] = (uint32_t)proj_repr_len(X, Y, Z, style) + 2;
}

__global__ void __launch_bounds__(FMT_BLOCK)
k_fmt_points_write(const uint32_t *__restrict__ proj, size_t n, fmt_point_style style,
                   const uint64_t *__restrict__ offs, char *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t X[8], Y[8], Z[8];
    fmt_ld8(X, proj + 24 * i);
    fmt_ld8(Y, proj + 24 * i + 8);
    fmt_ld8(Z, proj + 24 * i + 16);
    char *dst = out + offs[i];
    int o = proj_repr_write(X, Y, Z, style, dst);
    dst[o] = ',';
    dst[o + 1] = ' ';
}

__global__ void __launch_bounds__(FMT_BLOCK)
k_fmt_scalars_len(const uint32_t *__restrict__ sc, size_t n, int is_signed,
                  uint32_t *__restrict__ lens) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr a;
    fmt_ld8(a.v, sc + 8 * i);
    lens[i] = (uint32_t)fr_repr_len(a, is_signed != 0) + 2;
}

__global__ void __launch_bounds__(FMT_BLOCK)
k_fmt_scalars_write(const uint32_t *__restrict__ sc, size_t n, int is_signed,
                    const uint64_t *__restrict__ offs, char *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr a;
    fmt_ld8(a.v, sc + 8 * i);
    char *dst = out + offs[i];
    int o = fr_repr_write(a, is_signed != 0, dst);
    dst[o] = ',';
    dst[o + 1] = ' ';
}

static int fmt_common(vmpc_ctx *ctx, const void *src, size_t n, bool points, int is_signed,
                      void *out_text, size_t cap, uint64_t *len) {
    if (!ctx || !len || (n && (!src || !out_text))) return VMPC_E_INVAL;
    *len = 0;
    if (n == 0) return VMPC_OK;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    size_t need = vmpc_align(n * 4) + vmpc_align(n * 8) + vmpc_scan_ws_bytes(n, 8) + 512;
    VMPC_CHECK(vmpc_ws_reserve(ctx, need));
    uint32_t *lens = (uint32_t *)vmpc_ws_take(ctx, n * 4);
    uint64_t *offs = (uint64_t *)vmpc_ws_take(ctx, n * 8);
    uint64_t *total = (uint64_t *)vmpc_ws_take(ctx, 8);
    void *scan_ws = vmpc_ws_take(ctx, vmpc_scan_ws_bytes(n, 8));
    unsigned g = (unsigned)((n + FMT_BLOCK - 1) / FMT_BLOCK);
    {
        vmpc_stage_scope s(ctx, "format_len");
        if (points)
            k_fmt_points_len<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, g_point_style, lens);
        else
            k_fmt_scalars_len<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, is_signed, lens);
        VMPC_KERNEL_CHECK();
        VMPC_CHECK((vmpc_exclusive_scan<uint32_t, uint64_t>(st, lens, offs, n, scan_ws, total)));
    }
    uint64_t h_total = 0;
    VMPC_HIP_CHECK(hipMemcpyAsync(&h_total, total, 8, hipMemcpyDeviceToHost, st));
    VMPC_HIP_CHECK(hipStreamSynchronize(st));
    *len = h_total;
    if (h_total > cap) return VMPC_E_NOMEM;
    {
        vmpc_stage_scope s(ctx, "format_write");
        if (points)
            k_fmt_points_write<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, g_point_style, offs, (char *)out_text);
        else
            k_fmt_scalars_write<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, is_signed, offs,
                                                        (char *)out_text);
        VMPC_KERNEL_CHECK();
    }
    VMPC_HIP_CHECK(hipStreamSynchronize(st));
    return VMPC_OK;
}

// Asynchronous variant: everything (lengths, scan, text, the two device->host copies) is
// enqueued on the context's stream and the call returns at once.  `host_text` / `host_len` must
// be pinned (vmpc_host_alloc) and stay valid until the stream has been synchronised; `cap` bytes
// are copied (the text is at most ~1 % shorter than its worst case).  Lets the pre-image of the
// next Fiat-Shamir hash be produced and moved while the round's MSMs run.
// chunk_bytes > 0: the text goes to the host in pieces of that size, and the 32-bit word at host_len + 8 (pinned, 0
// when the call is made) counts the pieces that have landed - the host hashes piece k while piece k + 1 is on the
// link instead of waiting for all of it (a 2^20-point vector is 246 MB of text: 5 ms of link time, 100 ms of SHA-256).
static int fmt_async(vmpc_ctx *ctx, const void *src, size_t n, bool points, int is_signed, void *dev_text,
                     size_t cap, void *host_text, uint64_t *host_len, size_t chunk_bytes = 0) {
    if (!ctx || !host_len || (n && (!src || !dev_text || !host_text))) return VMPC_E_INVAL;
    size_t worst = n * (points ? (3 * 79 + 8) : (78 + 3));       // 78 digits + a sign per coordinate
    if (cap < worst) return VMPC_E_NOMEM;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (n == 0) {
        *host_len = 0;
        return VMPC_OK;
    }
    size_t need = vmpc_align(n * 4) + vmpc_align(n * 8) + vmpc_scan_ws_bytes(n, 8) + 512;
    VMPC_CHECK(vmpc_ws_reserve(ctx, need));
    uint32_t *lens = (uint32_t *)vmpc_ws_take(ctx, n * 4);
    uint64_t *offs = (uint64_t *)vmpc_ws_take(ctx, n * 8);
    uint64_t *total = (uint64_t *)vmpc_ws_take(ctx, 8);
    void *scan_ws = vmpc_ws_take(ctx, vmpc_scan_ws_bytes(n, 8));
    unsigned g = (unsigned)((n + FMT_BLOCK - 1) / FMT_BLOCK);
    vmpc_stage_scope s(ctx, "format_async");
    if (points)
        k_fmt_points_len<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, g_point_style, lens);
    else
        k_fmt_scalars_len<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, is_signed, lens);
    VMPC_KERNEL_CHECK();
    VMPC_CHECK((vmpc_exclusive_scan<uint32_t, uint64_t>(st, lens, offs, n, scan_ws, total)));
    if (points)
        k_fmt_points_write<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, g_point_style, offs, (char *)dev_text);
    else
        k_fmt_scalars_write<<<g, FMT_BLOCK, 0, st>>>((const uint32_t *)src, n, is_signed, offs,
                                                    (char *)dev_text);
    VMPC_KERNEL_CHECK();
    VMPC_HIP_CHECK(hipMemcpyAsync(host_len, total, 8, hipMemcpyDeviceToHost, st));
    if (chunk_bytes == 0) {
        VMPC_HIP_CHECK(hipMemcpyAsync(host_text, dev_text, worst, hipMemcpyDeviceToHost, st));
        return VMPC_OK;
    }
    void *landed_dev = nullptr;
    VMPC_HIP_CHECK(hipHostGetDevicePointer(&landed_dev, (char *)host_len + 8, 0));
    uint32_t k = 0;
    for (size_t off = 0; off < worst; off += chunk_bytes) {
        const size_t len = worst - off < chunk_bytes ? worst - off : chunk_bytes;
        VMPC_HIP_CHECK(hipMemcpyAsync((char *)host_text + off, (const char *)dev_text + off, len, hipMemcpyDeviceToHost, st));
        VMPC_HIP_CHECK(hipStreamWriteValue32(st, landed_dev, ++k, 0));
    }
    return VMPC_OK;
}

extern "C" int vmpc_format_points_async_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *dev_text,
                                            size_t cap, void *host_text, uint64_t *host_len) {
    return fmt_async(ctx, proj, n, true, 0, dev_text, cap, host_text, host_len);
}

extern "C" int vmpc_format_scalars_async_dev(vmpc_ctx *ctx, const void *scalars, size_t n, int is_signed,
                                             void *dev_text, size_t cap, void *host_text,
                                             uint64_t *host_len) {
    return fmt_async(ctx, scalars, n, false, is_signed, dev_text, cap, host_text, host_len);
}

extern "C" int vmpc_format_points_chunked_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *dev_text, size_t cap,
                                              void *host_text, uint64_t *host_len, size_t chunk_bytes) {
    if (chunk_bytes < 4096) return VMPC_E_INVAL;
    return fmt_async(ctx, proj, n, true, 0, dev_text, cap, host_text, host_len, chunk_bytes);
}

extern "C" int vmpc_format_scalars_chunked_dev(vmpc_ctx *ctx, const void *scalars, size_t n, int is_signed,
                                               void *dev_text, size_t cap, void *host_text, uint64_t *host_len,
                                               size_t chunk_bytes) {
    if (chunk_bytes < 4096) return VMPC_E_INVAL;
    return fmt_async(ctx, scalars, n, false, is_signed, dev_text, cap, host_text, host_len, chunk_bytes);
}

extern "C" int vmpc_host_alloc(size_t bytes, void **out) {
    if (!out) return VMPC_E_INVAL;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        snprintf(vmpc_err_buf, sizeof vmpc_err_buf, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return VMPC_E_NOMEM;
    }
    return VMPC_OK;
}

extern "C" int vmpc_host_free(void *p) {
    if (p) VMPC_HIP_CHECK(hipHostFree(p));
    return VMPC_OK;
}

extern "C" int vmpc_format_points_dev(vmpc_ctx *ctx, const void *proj, size_t n, void *out_text,
                                      size_t cap, uint64_t *len) {
    return fmt_common(ctx, proj, n, true, 0, out_text, cap, len);
}

extern "C" int vmpc_format_scalars_dev(vmpc_ctx *ctx, const void *scalars, size_t n, int is_signed,
                                       void *out_text, size_t cap, uint64_t *len) {
    return fmt_common(ctx, scalars, n, false, is_signed, out_text, cap, len);
}
