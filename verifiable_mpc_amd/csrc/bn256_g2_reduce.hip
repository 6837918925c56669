// BN-256 twist (F_p^2): bucket reduction
#include "bn256_impl.h"

template int bn_kernels<G2, BnF2>::reduce(vmpc_ctx *, const msm_plan &, msm_ws &);
