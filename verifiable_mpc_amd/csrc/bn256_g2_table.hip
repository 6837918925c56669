// BN-256 twist (F_p^2): fixed-base table build, fixed-base batch
#include "bn256_impl.h"

template int bn_kernels<G2, BnF2>::table_build(vmpc_ctx *, const void *, size_t, size_t, void *);
template int bn_kernels<G2, BnF2>::fixed_base(vmpc_ctx *, const void *, const void *, size_t, void *);
