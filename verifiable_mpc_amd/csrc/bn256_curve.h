// The curve half of the BN-256 MSM kernels as a policy type (csrc/bn256.hip, csrc/bn256_probe.hip): entries are
// affine points in Montgomery form, accumulators Jacobian; F = BnF1 (G1) or BnF2 (the sextic twist): fp29.h.
#pragma once
#include "sw256.h"
#include "fp29.h"

template <class F>
struct SwCurve {
    typedef aff<F> entry_t;
    typedef jac<F> acc_t;
    static constexpr int AFF_WORDS = 2 * F::WORDS;
    static constexpr int ENTRY_WORDS = 2 * F::WORDS;
    static constexpr int ACC_WORDS = 3 * F::WORDS;

    __device__ static entry_t entry_ld(const uint32_t *p) {
        entry_t e;
        e.x = F::load_raw(p);
        e.y = F::load_raw(p + F::WORDS);
        e.inf = F::is_zero(e.x) && F::is_zero(e.y);
        return e;
    }
    __device__ static void entry_st(uint32_t *p, const entry_t &e) {
        F::store_raw(p, e.inf ? F::zero() : e.x);
        F::store_raw(p + F::WORDS, e.inf ? F::zero() : e.y);
    }
    __device__ static acc_t acc_ld(const uint32_t *p) {
        acc_t a;
        a.X = F::load_raw(p);
        a.Y = F::load_raw(p + F::WORDS);
        a.Z = F::load_raw(p + 2 * F::WORDS);
        return a;
    }
    __device__ static void acc_st(uint32_t *p, const acc_t &a) {
        F::store_raw(p, a.X);
        F::store_raw(p + F::WORDS, a.Y);
        F::store_raw(p + 2 * F::WORDS, a.Z);
    }
    __device__ static acc_t identity() { return jac_identity<F>(); }
    __device__ static acc_t madd(const acc_t &a, entry_t e, bool neg) {
        e.y = F::select(e.y, F::neg(e.y), neg);
        return jac_madd<F>(a, e);
    }
};
// the fields the device kernels compute in (fp29.h)
typedef Fp29Ops BnF1;
typedef Fp29x2Ops BnF2;
typedef SwCurve<BnF1> G1;
typedef SwCurve<BnF2> G2;
