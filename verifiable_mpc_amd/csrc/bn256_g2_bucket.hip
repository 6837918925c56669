// BN-256 twist (F_p^2): entry preparation, bucket accumulation, split-bucket finish
#include "bn256_impl.h"

template int bn_kernels<G2, BnF2>::prep(vmpc_ctx *, const void *, size_t, uint32_t *);
template int bn_kernels<G2, BnF2>::bucket(vmpc_ctx *, const msm_plan &, msm_ws &, const uint32_t *);
