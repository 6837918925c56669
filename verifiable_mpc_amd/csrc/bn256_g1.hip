// BN-256 G1 (F_p): every kernel launcher of bn256_impl.h
#include "bn256_impl.h"

template int bn_kernels<G1, BnF1>::prep(vmpc_ctx *, const void *, size_t, uint32_t *);
template int bn_kernels<G1, BnF1>::bucket(vmpc_ctx *, const msm_plan &, msm_ws &, const uint32_t *);
template int bn_kernels<G1, BnF1>::reduce(vmpc_ctx *, const msm_plan &, msm_ws &);
template int bn_kernels<G1, BnF1>::final(vmpc_ctx *, const msm_plan &, msm_ws &, void *, void *);
template int bn_kernels<G1, BnF1>::final_multi(vmpc_ctx *, const msm_plan &, msm_ws &, void *, int);
template int bn_kernels<G1, BnF1>::table_build(vmpc_ctx *, const void *, size_t, size_t, void *);
template int bn_kernels<G1, BnF1>::validate(vmpc_ctx *, const void *, size_t, unsigned long long *);
template int bn_kernels<G1, BnF1>::fixed_base(vmpc_ctx *, const void *, const void *, size_t, void *);
