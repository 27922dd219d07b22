// K18: fused Adam over the flat trainable arena (depth decoder + pose decoder).
// Replaces torch.optim.Adam.step() at dpp.py:203,313 (defaults: betas (0.9, 0.999), eps 1e-8, no
// weight decay, no amsgrad) with the same operation order as torch's single-tensor CPU path:
//   m = m + (g - m)*(1-b1);  v = v*b2 + (1-b2)*g*g;
//   p = p - (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// HBM-bound: 4 streams read (p,g,m,v), 3 written, 16-byte accesses.
#include "adam_dev.h"

namespace clslam {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n4, size_t n, float step_size, float w1,
                                                   float beta2, float w2, float bc2_sqrt, float eps, float grad_scale,
                                                   const float* __restrict__ guard) {
    // guard (optional): the step's total loss.  NaN -> no update at all (dpp.py:1115-1118 raises before
    // backward/step; the host raises after this launch, see DepthPosePrediction.adapt).
    if (guard && !(guard[0] == guard[0])) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            adam_update(P[k], M[k], V[k], G[k] * grad_scale, step_size, w1, beta2, w2, bc2_sqrt, eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            float pk = p[i], mk = m[i], vk = v[i];
            adam_update(pk, mk, vk, g[i] * grad_scale, step_size, w1, beta2, w2, bc2_sqrt, eps);
            p[i] = pk; m[i] = mk; v[i] = vk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Up to kCopyBatch independent device-to-device copies in ONE launch (item table passed by value in the
// kernel arguments: no table upload).  blockIdx.y = item; 16-byte path when pointers and size allow it.
constexpr int kCopyBatch = 24;
struct CopyBatch {
    const void* src[kCopyBatch];
    void* dst[kCopyBatch];
    unsigned long long bytes[kCopyBatch];
};

__global__ __launch_bounds__(256) void copy_multi_kernel(CopyBatch b) {
    const int item = blockIdx.y;
    const unsigned long long bytes = b.bytes[item];
    const char* src = static_cast<const char*>(b.src[item]);
    char* dst = static_cast<char*>(b.dst[item]);
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if ((((unsigned long long)(uintptr_t)src | (unsigned long long)(uintptr_t)dst | bytes) & 15ull) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (size_t i = tid; i < bytes / 16; i += stride) d4[i] = s4[i];
    } else if ((((unsigned long long)(uintptr_t)src | (unsigned long long)(uintptr_t)dst | bytes) & 3ull) == 0) {
        const unsigned* s1 = reinterpret_cast<const unsigned*>(src);
        unsigned* d1 = reinterpret_cast<unsigned*>(dst);
        for (size_t i = tid; i < bytes / 4; i += stride) d1[i] = s1[i];
    } else {
        for (size_t i = tid; i < bytes; i += stride) dst[i] = src[i];
    }
}

}  // namespace clslam

using namespace clslam;

extern "C" int clslam_copy_multi(const clslam_copy_item* items, int nitems, void* stream) {
    CLSLAM_REQUIRE(nitems >= 0 && (items || nitems == 0), "copy_multi: bad args");
    for (int base = 0; base < nitems; base += kCopyBatch) {
        CopyBatch b;
        int n = 0;
        unsigned long long largest = 0;
        for (int i = base; i < nitems && n < kCopyBatch; ++i) {
            if (items[i].bytes == 0) continue;
            CLSLAM_REQUIRE(items[i].src && items[i].dst, "copy_multi: null pointer with a non-zero size");
            b.src[n] = items[i].src; b.dst[n] = items[i].dst; b.bytes[n] = items[i].bytes;
            largest = std::max<unsigned long long>(largest, items[i].bytes);
            ++n;
        }
        if (!n) continue;
        // 4 KiB per workgroup pass; small tables (a few 100 KiB per item) stay at a handful of workgroups per item
        const unsigned bx = (unsigned)std::min<unsigned long long>(128, std::max<unsigned long long>(1, (largest + 16383) / 16384));
        hipLaunchKernelGGL(copy_multi_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, b);
        const int rc = check_launch("copy_multi");
        if (rc != CLSLAM_OK) return rc;
    }
    return CLSLAM_OK;
}

extern "C" int clslam_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                                double beta1, double beta2, double eps, int step, float grad_scale, const float* guard, void* stream) {
    if (!n) return CLSLAM_OK;
    CLSLAM_REQUIRE(param && grad && exp_avg && exp_avg_sq && step >= 1, "adam_step: bad args");
    // scalars are formed in double like torch's python-side arithmetic, then rounded to fp32 once
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const size_t n4 = n / 4;
    const unsigned blocks = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (n4 + 255) / 256));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n4, n,
                       step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), bc2_sqrt, (float)eps, grad_scale, guard);
    return check_launch("adam_step");
}
