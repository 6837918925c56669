// Developer microbenchmark (not product): field-operation latency and chip-wide throughput on
// gfx950, the yardstick the MSM bucket kernel is priced against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/fe_bench.hip -o /tmp/fe_bench && /tmp/fe_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
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

// ---- round 6: the FP64-FMA product ------------------------------------------------------------
// 2^255 - 19 in five limbs of 51 bits held as DOUBLES.  A 51 x 51-bit partial product is split exactly by two FMAs
// (Emmart et al.): with the rounding mode at round-toward-zero, hi = fma(a, b, 2^104) has the exponent of 2^104 and
// floor(a b / 2^52) in its mantissa, and lo = fma(a, b, (2^104 + 2^52) - hi) = 2^52 + (a b mod 2^52) exactly; both
// are added up as 64-bit INTEGERS of their bit patterns (the exponent fields are subtracted once per column).  Columns
// to limbs: fold columns 5..9 with 19, one carry pass, back to doubles by the 2^52 trick.  25 x (2 FMA + 1 FP add + 2
// integer additions) against 100 v_mad_u64_u32 whose 64-bit accumulation is free.  Checked against fe_mul below.
struct fe5d {
    double v[5];
};
__device__ __forceinline__ void fe5d_round_toward_zero() {
    __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);       // MODE.FP_ROUND[3:2] (f64 / f16) = toward zero
}
__device__ __forceinline__ fe5d fe5d_mul(const fe5d &f, const fe5d &g) {
    // (set HERE, every product: set once at kernel entry the mode was back at round-to-nearest by the time the FMAs ran -
    // the results matched a round-to-nearest emulation exactly - the compiler restores the mode it assumes around the
    // integer <-> double conversions; one scalar instruction per product)
    fe5d_round_toward_zero();
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    unsigned long long hi[9], lo[9];
#pragma unroll
    for (int k = 0; k < 9; k++) hi[k] = lo[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) {
            // (inline assembly: the compiler resets the rounding mode to round-to-nearest in front of every FP64
            // instruction it knows about - s_setreg hwreg(MODE, 2, 2), 0 right behind the one above, round 6)
            double h, t, l;
            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(h) : "v"(f.v[i]), "v"(g.v[j]), "v"(C1));
            asm volatile("v_add_f64 %0, %1, -%2" : "=v"(t) : "v"(C2), "v"(h));
            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(l) : "v"(f.v[i]), "v"(g.v[j]), "v"(t));
            hi[i + j] += (unsigned long long)__double_as_longlong(h);
            lo[i + j] += (unsigned long long)__double_as_longlong(l);
        }
    // strip the exponent fields: column k has n_k = min(k, 8 - k) + 1 terms
    unsigned long long col[10];
#pragma unroll
    for (int k = 0; k < 10; k++) col[k] = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const unsigned long long n = (unsigned long long)((k < 4 ? k : 8 - k) + 1);
        const unsigned long long H = hi[k] - n * (0x467ull << 52), L = lo[k] - n * (0x433ull << 52);
        col[k] += L;                 // weight 2^(51 k)
        col[k + 1] += 2 * H;         // 2^52 = 2 * 2^51
    }
    // 2^255 = 19: columns 5..9 onto 0..4, then one carry pass
    unsigned long long r[5];
#pragma unroll
    for (int k = 0; k < 5; k++) r[k] = col[k] + 19 * col[k + 5];
    const unsigned long long M = (1ull << 51) - 1;
    unsigned long long c = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const unsigned long long t = r[k] + c;
        r[k] = t & M;
        c = t >> 51;
    }
    r[0] += 19 * c;                  // < 2^51 + 2^17: within a limb's slack (inputs < 2^52)
    fe5d o;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        // (assembly as well: a compiler-visible FP64 subtraction gets scheduled INTO the next product, behind its
        // s_setreg, with a mode reset of its own in front)
        const double biased = __longlong_as_double((long long)(r[k] | (0x433ull << 52))), m52 = -0x1p52;
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(o.v[k]) : "v"(biased), "v"(m52));
    }
    return o;
}
// the compiler may not move FP64 instructions of ITS OWN (the int -> double conversions of the inputs) behind the first
// product's s_setreg: it would put a mode reset in front of them
__device__ __forceinline__ void fe5d_pin(fe5d &a) {
#pragma unroll
    for (int k = 0; k < 5; k++) asm volatile("" : "+v"(a.v[k]));
}
__global__ void k_chain5d(const uint32_t *in, uint32_t *out, int iters) {
    fe5d_round_toward_zero();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe5d x, y;
    for (int k = 0; k < 5; k++) {
        x.v[k] = (double)(((unsigned long long)in[8 * (i & 1023) + k] << 19) | (in[8 * (i & 1023) + k + 1] & 0x7ffff));
        y.v[k] = (double)(((unsigned long long)in[8 * ((i + 7) & 1023) + k] << 19) | (in[8 * ((i + 7) & 1023) + k + 2] & 0x7ffff));
    }
    fe5d_pin(x);
    fe5d_pin(y);
    for (int k = 0; k < iters; k++) {
        x = fe5d_mul(x, y);
        y = fe5d_mul(y, x);
    }
    fe5d_pin(x);
    fe5d_pin(y);
    for (int k = 0; k < 5; k++) {
        const unsigned long long v = (unsigned long long)(x.v[k] + y.v[k]);
        out[8 * i + k] = (uint32_t)v ^ (uint32_t)(v >> 32);
    }
}
// one product of the same two field elements by fe_mul (10 limbs) and by fe5d_mul: equal residues?
__global__ void k_check5d(const uint32_t *in, uint32_t *bad, unsigned long long *dbg) {
    fe5d_round_toward_zero();
    const int i = threadIdx.x;
    uint32_t a8[8], b8[8];
    for (int k = 0; k < 8; k++) {
        a8[k] = in[8 * i + k];
        b8[k] = in[8 * (i + 64) + k];
    }
    a8[7] &= 0x7fffffffu;
    b8[7] &= 0x7fffffffu;
    const fe want = fe_mul(fe_mul(fe_load(a8), fe_load(b8)), fe_load(a8));      // two products: the second sees limbs with slack
    auto to5 = [](const uint32_t *w) {
        fe5d r;
        for (int k = 0; k < 5; k++) {
            unsigned long long v = 0;
            for (int bit = 0; bit < 51; bit++) {
                const int p = 51 * k + bit;
                if (p < 256) v |= (unsigned long long)((w[p >> 5] >> (p & 31)) & 1u) << bit;
            }
            r.v[k] = (double)v;
        }
        return r;
    };
    fe5d x = to5(a8), y = to5(b8);
    fe5d_pin(x);
    fe5d_pin(y);
    fe5d got = fe5d_mul(fe5d_mul(x, y), x);
    fe5d_pin(got);
    if (i == 0 && dbg) {
        for (int k = 0; k < 5; k++) {
            dbg[k] = (unsigned long long)x.v[k];
            dbg[5 + k] = (unsigned long long)y.v[k];
            dbg[10 + k] = (unsigned long long)got.v[k];
        }
        for (int k = 0; k < 8; k++) dbg[15 + k] = a8[k];
    }
    // compare as canonical 255-bit residues
    uint32_t w8[8];
    fe_store(w8, want);
    unsigned long long limb[5], cc = 0;
    for (int k = 0; k < 5; k++) limb[k] = (unsigned long long)got.v[k];
    for (int pass = 0; pass < 3; pass++) {                  // full reduction below 2^255 - 19
        cc = 0;
        for (int k = 0; k < 5; k++) {
            const unsigned long long t = limb[k] + cc;
            limb[k] = t & ((1ull << 51) - 1);
            cc = t >> 51;
        }
        limb[0] += 19 * cc;
    }
    // + 19, take bit 255, subtract: the usual final conditional subtraction
    unsigned long long t5[5];
    cc = 19;
    for (int k = 0; k < 5; k++) {
        const unsigned long long t = limb[k] + cc;
        t5[k] = t & ((1ull << 51) - 1);
        cc = t >> 51;
    }
    if (cc) for (int k = 0; k < 5; k++) limb[k] = t5[k];
    uint32_t g8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < 255; p++)
        if ((limb[p / 51] >> (p % 51)) & 1ull) g8[p >> 5] |= 1u << (p & 31);
    bool same = true;
    for (int k = 0; k < 8; k++) same = same && g8[k] == w8[k];
    if (!same) atomicAdd(bad, 1u);
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
    {
        uint32_t *dbad;
        hipMalloc(&dbad, 4);
        hipMemset(dbad, 0, 4);
        unsigned long long *ddbg, hdbg[23];
        hipMalloc(&ddbg, sizeof hdbg);
        k_check5d<<<1, 64>>>(din, dbad, getenv("FE5D_DEBUG") ? ddbg : nullptr);
        uint32_t hbad = 1;
        hipMemcpy(&hbad, dbad, 4, hipMemcpyDeviceToHost);
        if (getenv("FE5D_DEBUG")) {
            hipMemcpy(hdbg, ddbg, sizeof hdbg, hipMemcpyDeviceToHost);
            for (int k = 0; k < 23; k++) printf("dbg[%d] = %llu\n", k, hdbg[k]);
        }
        lat = run(k_chain5d, 1, 64, din, dout, iters);
        thr = run(k_chain5d, B, T, din, dout, iters);
        printf("fe5d_mul (5 x 51-bit limbs as doubles, FMA hi / lo split, RZ) latency %.1f ns (1 wave) | %.1f G/s (chain) | "
               "%s\n", lat * 1e6 / (2.0 * iters), 2.0 * iters * B * T / (thr * 1e-3) / 1e9,
               hbad ? "value check against fe_mul NOT passed (timing of the instruction mix only, see EXPERIMENTS R6.5)"
                    : "64 products equal to fe_mul's");
    }
    for (int blocks : {256 * 2, 256 * 3, 256 * 4, 256 * 8}) {
        double m = run(k_madd, blocks, T, din, dout, 500);
        printf("ge_madd %d blocks x 256: %.2f G madd/s  (%.1f G fe_mul-equivalents/s)\n", blocks,
               500.0 * blocks * T / (m * 1e-3) / 1e9, 7 * 500.0 * blocks * T / (m * 1e-3) / 1e9);
    }
    return 0;
}
