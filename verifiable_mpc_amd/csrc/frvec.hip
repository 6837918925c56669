// Scalar-field (mod l) vector kernels of Protocol 4/5's scalar side:
//   vmpc_fr_axpy_dev   out = c*x + y      z' = z_l + c*z_r, L' = c*L_l + L_r (compressed_pivot.py:70-76),
//                                         z = c0*x + r (:134)
//   vmpc_fr_scale_dev  out = c*x          L~ = (L||0)*c1 (:141)
//   vmpc_fr_dot_dev    sum a_i*b_i        LinearForm evaluation (pivot.py:84-92), L~(0||z_l), L~(z_r||0)
// HBM-bound streaming kernels: 32-B elements, two 16-B accesses per lane, grid capped at
// 2048 workgroups with a grid-stride loop.
#include "common.h"
#include "fr.h"

#define FR_BLOCK 256
#define FR_MAX_GRID 2048

struct fr_arg {
    uint32_t v[8];
};

__device__ __forceinline__ fr frv_ld(const uint32_t *src) {
    const uint4 *p = reinterpret_cast<const uint4 *>(src);
    uint4 a = p[0], b = p[1];
    fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void frv_st(uint32_t *dst, const fr &a) {
    uint4 *p = reinterpret_cast<uint4 *>(dst);
    p[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    p[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

__global__ void __launch_bounds__(FR_BLOCK)
k_fr_axpy(fr_arg c, const uint32_t *__restrict__ x, const uint32_t *__restrict__ y, size_t n,
          uint32_t *__restrict__ out) {
    fr cc;
#pragma unroll
    for (int i = 0; i < 8; i++) cc.v[i] = c.v[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        fr r = fr_mul(cc, frv_ld(x + 8 * i));
        if (y) r = fr_add(r, frv_ld(y + 8 * i));
        frv_st(out + 8 * i, r);
    }
}

// the same with one more element behind the n results (z_hat = (c0 x + r) || phi, L~ = c1 (L || 0):
// compressed_pivot.py:134-141) - no copy of the vector to append a scalar
__global__ void __launch_bounds__(FR_BLOCK)
k_fr_axpy_tail(fr_arg c, const uint32_t *__restrict__ x, const uint32_t *__restrict__ y, size_t n, fr_arg tail,
               uint32_t *__restrict__ out) {
    fr cc;
#pragma unroll
    for (int i = 0; i < 8; i++) cc.v[i] = c.v[i];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        fr r = fr_mul(cc, frv_ld(x + 8 * i));
        if (y) r = fr_add(r, frv_ld(y + 8 * i));
        frv_st(out + 8 * i, r);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        fr t;
#pragma unroll
        for (int i = 0; i < 8; i++) t.v[i] = tail.v[i];
        frv_st(out + 8 * n, t);
    }
}

__global__ void __launch_bounds__(FR_BLOCK)
k_fr_dot(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n,
         uint32_t *__restrict__ partials) {
    __shared__ uint32_t lds[FR_BLOCK * 8];
    fr acc = fr_zero();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        acc = fr_add(acc, fr_mul(frv_ld(a + 8 * i), frv_ld(b + 8 * i)));
    frv_st(lds + 8 * threadIdx.x, acc);
    __syncthreads();
    for (int stride = FR_BLOCK / 2; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride)
            frv_st(lds + 8 * threadIdx.x,
                   fr_add(frv_ld(lds + 8 * threadIdx.x), frv_ld(lds + 8 * (threadIdx.x + stride))));
        __syncthreads();
    }
    if (threadIdx.x == 0) frv_st(partials + 8 * blockIdx.x, frv_ld(lds));
}

__global__ void __launch_bounds__(FR_BLOCK)
k_fr_sum(const uint32_t *__restrict__ v, size_t n, uint32_t *__restrict__ out) {
    __shared__ uint32_t lds[FR_BLOCK * 8];
    fr acc = fr_zero();
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) acc = fr_add(acc, frv_ld(v + 8 * i));
    frv_st(lds + 8 * threadIdx.x, acc);
    __syncthreads();
    for (int stride = FR_BLOCK / 2; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x < stride)
            frv_st(lds + 8 * threadIdx.x,
                   fr_add(frv_ld(lds + 8 * threadIdx.x), frv_ld(lds + 8 * (threadIdx.x + stride))));
        __syncthreads();
    }
    if (threadIdx.x == 0) frv_st(out, frv_ld(lds));
}

__global__ void __launch_bounds__(FR_BLOCK)
k_fr_check(const uint32_t *__restrict__ v, size_t n, uint32_t *__restrict__ status) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        fr a = frv_ld(v + 8 * i);
        if (fr_geq_l(a.v)) atomicAdd(&status[VMPC_ST_NONCANON], 1u);
    }
}

// v[j] = z[j mod 2^low_bits] * prod_{i<R} (c_i if bit (low_bits + R - 1 - i) of j is 0 else 1)
// The generator fold g' = c*g_l + g_r (compressed_pivot.py:64) applied R times to a vector of
// 2^(R+low_bits) elements is the linear map  g_final[t] = sum_{j = t mod 2^low_bits} s[j] * g[j]
// with these coefficients: round i multiplies the LEFT half (top remaining index bit 0) by c_i.
// Lets a verifier (or the prover's short tail) replace R element-wise folds by one MSM.
struct fr_chal_arg {
    uint32_t c[20][8];
};
__global__ void __launch_bounds__(FR_BLOCK)
k_fr_challenge_products(fr_chal_arg ch, int R, int low_bits, const uint32_t *__restrict__ z,
                        size_t n, uint32_t *__restrict__ out) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (size_t)gridDim.x * blockDim.x) {
        fr acc = frv_ld(z + 8 * (j & (((size_t)1 << low_bits) - 1)));
        for (int i = 0; i < R; i++) {
            int bit = low_bits + R - 1 - i;
            if (((j >> bit) & 1) == 0) {
                fr c;
#pragma unroll
                for (int k = 0; k < 8; k++) c.v[k] = ch.c[i][k];
                acc = fr_mul(acc, c);
            }
        }
        frv_st(out + 8 * j, acc);
    }
}

// Scalars of A_i and B_i (compressed_pivot.py:41-42) over a FIXED base vector g0 of 2^log2_m0
// generators when the last t folds have NOT been applied to the generators: with
// s[j] = prod_{r<t} (c_r if bit (log2_m0-1-r) of j is 0 else 1), m = 2^log2_m0 / 2^t, h = m/2,
//   A = sum_j [ (j mod m) >= h ] * z[(j mod m) - h] * s[j] * g0[j]
//   B = sum_j [ (j mod m) <  h ] * z[(j mod m) + h] * s[j] * g0[j]
// (zeros are skipped by the MSM's digit sort).  Lets the short tail of Protocol 4 skip the
// element-wise generator fold, whose 253-doubling chain is pure latency there.
__global__ void __launch_bounds__(FR_BLOCK)
k_fr_tail_scalars(fr_chal_arg ch, int t, int log2_m0, const uint32_t *__restrict__ z,
                  uint32_t *__restrict__ out_a, uint32_t *__restrict__ out_b) {
    const size_t m0 = (size_t)1 << log2_m0;
    const size_t m = m0 >> t, h = m >> 1;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m0;
         j += (size_t)gridDim.x * blockDim.x) {
        size_t u = j & (m - 1);
        bool right = u >= h;
        fr acc = frv_ld(z + 8 * (right ? u - h : u + h));
        for (int r = 0; r < t; r++) {
            if (((j >> (log2_m0 - 1 - r)) & 1) == 0) {
                fr c;
#pragma unroll
                for (int k = 0; k < 8; k++) c.v[k] = ch.c[r][k];
                acc = fr_mul(acc, c);
            }
        }
        frv_st(out_a + 8 * j, right ? acc : fr_zero());
        frv_st(out_b + 8 * j, right ? fr_zero() : acc);
    }
}

// incremental form: `prod` holds s[j] for the first t-1 challenges (all ones for t = 0) and is updated
// in place with the newest one, so a round costs two products per element instead of up to t + 1
__global__ void __launch_bounds__(FR_BLOCK)
k_fr_tail_scalars_inc(fr_arg c_new, const uint32_t *__restrict__ c_mem, int t, int log2_m0,
                      const uint32_t *__restrict__ z, size_t j0, size_t count,
                      uint32_t *__restrict__ prod, uint32_t *__restrict__ out_a, uint32_t *__restrict__ out_b) {
    // positions j0 .. j0 + count - 1 of the length-2^log2_m0 vectors; prod / out_a / out_b hold that block only
    // (a rank of the sharded prover owns one block of g_hat: verifiable_mpc_amd/sharded.py)
    const size_t m0 = (size_t)1 << log2_m0;
    const size_t m = m0 >> t, h = m >> 1;
    // c_mem: the challenge was not known when the launch was queued (prover.hip: rounds queued behind a stream
    // wait) - it is read from device-visible memory instead of the argument
    fr c;
#pragma unroll
    for (int k = 0; k < 8; k++) c.v[k] = c_mem ? c_mem[k] : c_new.v[k];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = j0 + i;
        fr s;
        if (t == 0) {
            s = fr_zero();
            s.v[0] = 1;
        } else {
            s = frv_ld(prod + 8 * i);
            if (((j >> (log2_m0 - t)) & 1) == 0) s = fr_mul(s, c);
        }
        frv_st(prod + 8 * i, s);
        size_t u = j & (m - 1);
        bool right = u >= h;
        fr acc = fr_mul(frv_ld(z + 8 * (right ? u - h : u + h)), s);
        frv_st(out_a + 8 * i, right ? acc : fr_zero());
        frv_st(out_b + 8 * i, right ? fr_zero() : acc);
    }
}

static inline unsigned fr_grid(size_t n) {
    size_t g = (n + FR_BLOCK - 1) / FR_BLOCK;
    return (unsigned)(g > FR_MAX_GRID ? FR_MAX_GRID : (g ? g : 1));
}

static int fr_arg_from(const uint8_t c[32], fr_arg &a) {
    memcpy(a.v, c, 32);
    return fr_geq_l(a.v) ? VMPC_E_NONCANON : VMPC_OK;
}

extern "C" int vmpc_fr_axpy_dev(vmpc_ctx *ctx, const uint8_t c[32], const void *x, const void *y,
                                size_t n, void *out) {
    if (!ctx || !c || (n && (!x || !out))) return VMPC_E_INVAL;
    if (n == 0) return VMPC_OK;
    fr_arg ca;
    VMPC_CHECK(fr_arg_from(c, ca));
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "fr_axpy");
    k_fr_axpy<<<fr_grid(n), FR_BLOCK, 0, ctx->stream>>>(ca, (const uint32_t *)x, (const uint32_t *)y, n,
                                                        (uint32_t *)out);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_fr_axpy_tail_dev(vmpc_ctx *ctx, const uint8_t c[32], const void *x, const void *y, size_t n,
                                     const uint8_t tail[32], void *out) {
    if (!ctx || !c || !tail || !out || (n && !x)) return VMPC_E_INVAL;
    fr_arg ca, ta;
    VMPC_CHECK(fr_arg_from(c, ca));
    VMPC_CHECK(fr_arg_from(tail, ta));
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "fr_axpy");
    k_fr_axpy_tail<<<n ? fr_grid(n) : 1, FR_BLOCK, 0, ctx->stream>>>(ca, (const uint32_t *)x, (const uint32_t *)y, n, ta,
                                                                      (uint32_t *)out);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_fr_scale_dev(vmpc_ctx *ctx, const uint8_t c[32], const void *x, size_t n,
                                 void *out) {
    return vmpc_fr_axpy_dev(ctx, c, x, nullptr, n, out);
}

extern "C" int vmpc_fr_dot_dev(vmpc_ctx *ctx, const void *a, const void *b, size_t n,
                               uint8_t out[32]) {
    if (!ctx || !out || (n && (!a || !b))) return VMPC_E_INVAL;
    if (n == 0) {
        memset(out, 0, 32);
        return VMPC_OK;
    }
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    unsigned g = fr_grid(n);
    VMPC_CHECK(vmpc_ws_reserve(ctx, vmpc_align((size_t)g * 32) + 256));
    uint32_t *partials = (uint32_t *)vmpc_ws_take(ctx, (size_t)g * 32);
    uint32_t *res = (uint32_t *)vmpc_ws_take(ctx, 32);
    {
        vmpc_stage_scope s(ctx, "fr_dot");
        k_fr_dot<<<g, FR_BLOCK, 0, ctx->stream>>>((const uint32_t *)a, (const uint32_t *)b, n, partials);
        VMPC_KERNEL_CHECK();
        k_fr_sum<<<1, FR_BLOCK, 0, ctx->stream>>>(partials, g, res);
        VMPC_KERNEL_CHECK();
    }
    VMPC_HIP_CHECK(hipMemcpyAsync(out, res, 32, hipMemcpyDeviceToHost, ctx->stream));
    VMPC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VMPC_OK;
}

// the same inner product left on the device (no host round trip): out_dev receives 32 bytes when the
// stream reaches it - the exponent of k in A_i / B_i (compressed_pivot.py:41-42) goes straight into the
// commitment's extra-scalar buffer
extern "C" int vmpc_fr_dot_to_dev(vmpc_ctx *ctx, const void *a, const void *b, size_t n, void *out_dev) {
    if (!ctx || !out_dev || (n && (!a || !b))) return VMPC_E_INVAL;
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) {
        VMPC_HIP_CHECK(hipMemsetAsync(out_dev, 0, 32, ctx->stream));
        return VMPC_OK;
    }
    unsigned g = fr_grid(n);
    VMPC_CHECK(vmpc_ws_reserve(ctx, vmpc_align((size_t)g * 32) + 256));
    uint32_t *partials = (uint32_t *)vmpc_ws_take(ctx, (size_t)g * 32);
    vmpc_stage_scope s(ctx, "fr_dot");
    k_fr_dot<<<g, FR_BLOCK, 0, ctx->stream>>>((const uint32_t *)a, (const uint32_t *)b, n, partials);
    VMPC_KERNEL_CHECK();
    k_fr_sum<<<1, FR_BLOCK, 0, ctx->stream>>>(partials, g, (uint32_t *)out_dev);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_fr_challenge_products_dev(vmpc_ctx *ctx, const uint8_t *challenges, int rounds,
                                              int low_bits, const void *z, size_t n, void *out) {
    if (!ctx || rounds < 0 || rounds > 20 || low_bits < 0 || low_bits > 40 || (rounds && !challenges) ||
        (n && (!z || !out)))
        return VMPC_E_INVAL;
    if (n != ((size_t)1 << (rounds + low_bits))) return VMPC_E_INVAL;
    fr_chal_arg a;
    memset(&a, 0, sizeof a);
    for (int i = 0; i < rounds; i++) {
        memcpy(a.c[i], challenges + 32 * i, 32);
        if (fr_geq_l(a.c[i])) return VMPC_E_NONCANON;
    }
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "fr_challenge_products");
    k_fr_challenge_products<<<fr_grid(n), FR_BLOCK, 0, ctx->stream>>>(a, rounds, low_bits,
                                                                       (const uint32_t *)z, n,
                                                                       (uint32_t *)out);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_fr_tail_scalars_dev(vmpc_ctx *ctx, const uint8_t *challenges, int t, int log2_m0,
                                        const void *z, void *out_a, void *out_b) {
    if (!ctx || t < 0 || t > 20 || log2_m0 < 1 || log2_m0 > 40 || t >= log2_m0 || (t && !challenges) ||
        !z || !out_a || !out_b)
        return VMPC_E_INVAL;
    fr_chal_arg a;
    memset(&a, 0, sizeof a);
    for (int i = 0; i < t; i++) {
        memcpy(a.c[i], challenges + 32 * i, 32);
        if (fr_geq_l(a.c[i])) return VMPC_E_NONCANON;
    }
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "fr_tail_scalars");
    k_fr_tail_scalars<<<fr_grid((size_t)1 << log2_m0), FR_BLOCK, 0, ctx->stream>>>(
        a, t, log2_m0, (const uint32_t *)z, (uint32_t *)out_a, (uint32_t *)out_b);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_fr_tail_scalars_inc_dev(vmpc_ctx *ctx, const uint8_t newest_challenge[32], int t, int log2_m0,
                                            const void *z, void *products, void *out_a, void *out_b) {
    return vmpc_fr_tail_scalars_block_dev(ctx, newest_challenge, t, log2_m0, z, 0, (size_t)1 << (log2_m0 > 40 ? 0 : log2_m0),
                                          products, out_a, out_b);
}

// challenge_mem != NULL: the newest challenge is read by the kernel from that device-visible address (8 words, a
// canonical residue) - for launches queued before the challenge exists (prover.hip)
int vmpc_fr_tail_scalars_block_mem(vmpc_ctx *ctx, const uint8_t newest_challenge[32], const uint32_t *challenge_mem,
                                   int t, int log2_m0, const void *z, size_t j0, size_t count, void *products,
                                   void *out_a, void *out_b) {
    if (!ctx || t < 0 || t > 40 || log2_m0 < 1 || log2_m0 > 40 || t >= log2_m0 ||
        (t && !newest_challenge && !challenge_mem) || !z || !products || !out_a || !out_b ||
        j0 + count > ((size_t)1 << log2_m0))
        return VMPC_E_INVAL;
    if (count == 0) return VMPC_OK;
    fr_arg a;
    memset(&a, 0, sizeof a);
    if (t && !challenge_mem) VMPC_CHECK(fr_arg_from(newest_challenge, a));
    VMPC_HIP_CHECK(hipSetDevice(ctx->device));
    vmpc_stage_scope s(ctx, "fr_tail_scalars");
    k_fr_tail_scalars_inc<<<fr_grid(count), FR_BLOCK, 0, ctx->stream>>>(
        a, t ? challenge_mem : nullptr, t, log2_m0, (const uint32_t *)z, j0, count, (uint32_t *)products,
        (uint32_t *)out_a, (uint32_t *)out_b);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}

extern "C" int vmpc_fr_tail_scalars_block_dev(vmpc_ctx *ctx, const uint8_t newest_challenge[32], int t, int log2_m0,
                                              const void *z, size_t j0, size_t count, void *products, void *out_a,
                                              void *out_b) {
    return vmpc_fr_tail_scalars_block_mem(ctx, newest_challenge, nullptr, t, log2_m0, z, j0, count, products, out_a,
                                          out_b);
}

// canonical-residue check of a scalar vector (C-ABI boundary hygiene, SURVEY.md hard part 5)
int vmpc_fr_check_dev(vmpc_ctx *ctx, const void *v, size_t n) {
    if (n == 0) return VMPC_OK;
    k_fr_check<<<fr_grid(n), FR_BLOCK, 0, ctx->stream>>>((const uint32_t *)v, n, ctx->d_status);
    VMPC_KERNEL_CHECK();
    return VMPC_OK;
}
