// BN-256 twist (F_p^2): recombination, validation
#include "bn256_impl.h"

template int bn_kernels<G2, BnF2>::final(vmpc_ctx *, const msm_plan &, msm_ws &, void *, void *);
template int bn_kernels<G2, BnF2>::final_multi(vmpc_ctx *, const msm_plan &, msm_ws &, void *, int);
template int bn_kernels<G2, BnF2>::validate(vmpc_ctx *, const void *, size_t, unsigned long long *);
