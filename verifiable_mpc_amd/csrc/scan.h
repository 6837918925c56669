// Device-wide exclusive prefix sum of uint32 counts (wave64 shuffles + LDS), used for
// bucket start offsets of the Pippenger sort and for text offsets of the transcript
// formatter.  Three-kernel reduce-then-scan; inputs of up to 2048 * 2048 * 2048 items.
#pragma once
#include "common.h"

#define VMPC_SCAN_THREADS 256
#define VMPC_SCAN_ITEMS 8
#define VMPC_SCAN_TILE (VMPC_SCAN_THREADS * VMPC_SCAN_ITEMS)

template <typename T>
__device__ __forceinline__ T vmpc_wave_incl_scan(T v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = block sum
template <typename T>
__device__ __forceinline__ T vmpc_block_excl_scan(T v, T *total, T *lds /* >= 4 entries */) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = vmpc_wave_incl_scan(v, lane);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    T wave_off = 0, tot = 0;
    int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; w++) {
        T s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + incl - v;
}

template <typename InT, typename OutT>
__global__ void __launch_bounds__(VMPC_SCAN_THREADS)
k_scan_tiles(const InT *__restrict__ in, OutT *__restrict__ out, OutT *__restrict__ tile_sums,
             size_t n) {
    __shared__ OutT lds[8];
    size_t base = (size_t)blockIdx.x * VMPC_SCAN_TILE + (size_t)threadIdx.x * VMPC_SCAN_ITEMS;
    OutT v[VMPC_SCAN_ITEMS];
    OutT s = 0;
    // a thread's eight 4-byte items as two 16-byte vectors (one request per line instead of eight 32-byte-strided
    // ones): the 1.7 M chunk counters of a wide-window commitment took 15 us per pass item by item
    const bool vec = sizeof(InT) == 4 && sizeof(OutT) == 4 && VMPC_SCAN_ITEMS == 8 && base + VMPC_SCAN_ITEMS <= n &&
                     (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (vec) {
        const uint4 a = reinterpret_cast<const uint4 *>(in + base)[0], b = reinterpret_cast<const uint4 *>(in + base)[1];
        v[0] = (OutT)a.x; v[1] = (OutT)a.y; v[2] = (OutT)a.z; v[3] = (OutT)a.w;
        v[4] = (OutT)b.x; v[5] = (OutT)b.y; v[6] = (OutT)b.z; v[7] = (OutT)b.w;
#pragma unroll
        for (int i = 0; i < VMPC_SCAN_ITEMS; i++) s += v[i];
    } else {
#pragma unroll
        for (int i = 0; i < VMPC_SCAN_ITEMS; i++) {
            v[i] = (base + i < n) ? (OutT)in[base + i] : (OutT)0;
            s += v[i];
        }
    }
    OutT tot;
    OutT ex = vmpc_block_excl_scan<OutT>(s, &tot, lds);
    if (vec) {
        uint32_t e[VMPC_SCAN_ITEMS];
#pragma unroll
        for (int i = 0; i < VMPC_SCAN_ITEMS; i++) {
            e[i] = (uint32_t)ex;
            ex += v[i];
        }
        reinterpret_cast<uint4 *>(out + base)[0] = make_uint4(e[0], e[1], e[2], e[3]);
        reinterpret_cast<uint4 *>(out + base)[1] = make_uint4(e[4], e[5], e[6], e[7]);
    } else {
#pragma unroll
        for (int i = 0; i < VMPC_SCAN_ITEMS; i++) {
            if (base + i < n) out[base + i] = ex;
            ex += v[i];
        }
    }
    if (threadIdx.x == 0 && tile_sums) tile_sums[blockIdx.x] = tot;
}

template <typename OutT>
__global__ void __launch_bounds__(VMPC_SCAN_THREADS)
k_scan_add(OutT *__restrict__ out, const OutT *__restrict__ tile_offsets, size_t n) {
    size_t base = (size_t)blockIdx.x * VMPC_SCAN_TILE + (size_t)threadIdx.x * VMPC_SCAN_ITEMS;
    OutT off = tile_offsets[blockIdx.x];
    if (sizeof(OutT) == 4 && VMPC_SCAN_ITEMS == 8 && base + VMPC_SCAN_ITEMS <= n && ((uintptr_t)out & 15) == 0) {
        uint4 *p = reinterpret_cast<uint4 *>(out + base);
        uint4 a = p[0], b = p[1];
        const uint32_t o = (uint32_t)off;
        a.x += o; a.y += o; a.z += o; a.w += o;
        b.x += o; b.y += o; b.z += o; b.w += o;
        p[0] = a;
        p[1] = b;
        return;
    }
#pragma unroll
    for (int i = 0; i < VMPC_SCAN_ITEMS; i++)
        if (base + i < n) out[base + i] += off;
}

// up to 32768 items in ONE workgroup of 1024 threads (the histogram of a short MSM - 16640 counters in a late
// prover round: three launches cost three dispatches)
#define VMPC_SCAN_SMALL_MAX 32768
// ITEMS = 8 for up to 8192 items: a 1024-thread workgroup is sixteen waves on ONE CU, and with 32 items in registers
// they need (nearly) a whole register file - beside a latency-bound kernel that keeps one wave on every SIMD of the
// chip (the exact fold of a short vector, csrc/exact.hip) the workgroup waited for that kernel to END: 785 us for the
// text offsets of the fold's previous slice (profiles/r06_fold_slices_timeline.txt).  Eight items fit beside it.
template <typename InT, typename OutT, int ITEMS>
__global__ void __launch_bounds__(1024)
k_scan_small(const InT *__restrict__ in, OutT *__restrict__ out, OutT *__restrict__ total_out, size_t n) {
    __shared__ OutT lds[16];
    const int per = (int)((n + 1023) / 1024);                  // <= ITEMS consecutive items per thread
    const size_t base = (size_t)threadIdx.x * per;
    OutT v[ITEMS];
    OutT s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        v[i] = (i < per && base + i < n) ? (OutT)in[base + i] : (OutT)0;
        s += v[i];
    }
    OutT tot;
    OutT ex = vmpc_block_excl_scan<OutT>(s, &tot, lds);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        if (i < per && base + i < n) out[base + i] = ex;
        ex += v[i];
    }
    if (threadIdx.x == 0 && total_out) *total_out = tot;
}

inline size_t vmpc_scan_ws_bytes(size_t n, size_t elem) {
    size_t total = 0;
    while (n > 1) {
        size_t tiles = (n + VMPC_SCAN_TILE - 1) / VMPC_SCAN_TILE;
        total += vmpc_align(tiles * elem);
        n = tiles;
        if (tiles == 1) break;
    }
    return total + 256;
}

// out[i] = sum_{j<i} in[j]; if total_out != nullptr, *total_out (device) = sum of all.
// `ws` must hold vmpc_scan_ws_bytes(n, sizeof(OutT)).
template <typename InT, typename OutT>
int vmpc_exclusive_scan(hipStream_t stream, const InT *in, OutT *out, size_t n, void *ws,
                        OutT *total_out) {
    if (n == 0) {
        if (total_out) VMPC_HIP_CHECK(hipMemsetAsync(total_out, 0, sizeof(OutT), stream));
        return VMPC_OK;
    }
    size_t tiles = (n + VMPC_SCAN_TILE - 1) / VMPC_SCAN_TILE;
    OutT *sums = (OutT *)ws;
    char *next_ws = (char *)ws + vmpc_align(tiles * sizeof(OutT));
    if (tiles > 1 && n <= VMPC_SCAN_SMALL_MAX) {
        if (n <= 8192) k_scan_small<InT, OutT, 8><<<1, 1024, 0, stream>>>(in, out, total_out, n);
        else k_scan_small<InT, OutT, 32><<<1, 1024, 0, stream>>>(in, out, total_out, n);
        VMPC_KERNEL_CHECK();
        return VMPC_OK;
    }
    if (tiles == 1) {
        k_scan_tiles<InT, OutT><<<1, VMPC_SCAN_THREADS, 0, stream>>>(in, out, total_out, n);
        VMPC_KERNEL_CHECK();
        return VMPC_OK;
    }
    k_scan_tiles<InT, OutT><<<(unsigned)tiles, VMPC_SCAN_THREADS, 0, stream>>>(in, out, sums, n);
    VMPC_KERNEL_CHECK();
    // scan the tile sums in place (recursively), then add them back
    VMPC_CHECK((vmpc_exclusive_scan<OutT, OutT>(stream, sums, sums, tiles, next_ws, total_out)));
    k_scan_add<OutT><<<(unsigned)tiles, VMPC_SCAN_THREADS, 0, stream>>>(out, sums, n);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}
