// ELU's negative branch for the fp32 training kernels (train.hip, tft.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mcmc {
namespace {

// expm1 for the ELU's negative branch: exp(z) - 1 where that keeps its relative accuracy (|z| >= 1/16: within 1e-6 of expm1),
// the degree-4 Taylor polynomial below it (truncation z^4/120 < 1.3e-7 relative).  One v_exp_f32 and six FMAs / selects per
// element; Kahan's (u - 1) * z / log(u) form used before (2 ulp) cost three transcendentals and ~20 % of the large-batch forward pass.
__device__ __forceinline__ float elu_neg(float z) {
    const float big = __expf(z) - 1.0f;
    const float small = z * fmaf(z, fmaf(z, fmaf(z, 1.0f / 24.0f, 1.0f / 6.0f), 0.5f), 1.0f);
    return z > -0.0625f ? small : big;
}

}  // namespace
}  // namespace mcmc
