// The Adam update of one element, shared by adam_kernel (adam.hip) and the fused reduction + update (conv_bwd.hip) so that
// both produce the same bits: floating-point contraction is switched off here (whether `v * beta2 + w2 * g * g` becomes an
// fma is otherwise the compiler's choice per call site).  Operation order of torch.optim.Adam's single-tensor path
// (dpp.py:203,313): m = m + (g - m)(1 - b1); v = v b2 + (1 - b2) g g; p = p - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
#pragma once
#include "common.h"

namespace clslam {

__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float step_size, float w1, float beta2,
                                            float w2, float bc2_sqrt, float eps) {
#pragma clang fp contract(off)
    m = m + (g - m) * w1;
    v = v * beta2 + w2 * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

}  // namespace clslam
