// Image-pyramid ingest: the LANCZOS resize + ToTensor of the reference's datasets
// (datasets/utils.py:62-66 Resize(LANCZOS) per scale, :154-163 every level from the previous one,
// :213-215 ToTensor) on uint8 images that were uploaded once, instead of PIL on one host thread.
//
// torchvision's Resize on a PIL image is PIL.Image.resize(size, LANCZOS) = Pillow's ImagingResample 8-bit
// path (libImaging/Resample.c; not part of /root/reference, restated in oracle/resize.py): per output
// index a window of taps with double-precision Lanczos-3 weights normalised to 1, converted to 22-bit
// fixed point; horizontal pass then vertical pass, int32 accumulation from 1 << 21, (acc >> 22) clipped
// to [0,255], uint8 intermediate.  The tap plan is computed on the HOST with libm exactly like Pillow
// (clslam_lanczos_plan); the kernels are integer multiply-accumulates -> results are BIT-EXACT.
// HBM-bound byte work: 3 B read + 3 B written per pixel per pass (+12 B for the fused float planes).
#include <cmath>
#include <vector>

#include "common.h"

// Pillow's C is compiled without fused multiply-add: every rounding below must happen where C puts it
// (hipcc contracts a*b+c by default, and __fmul_rn/__fadd_rn are plain operators in the ROCm headers).
#pragma clang fp contract(off)

namespace clslam {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;

// One thread per output pixel (all C <= 4 channels).  axis 1: out (B,H,out,C) from in (B,H,in_w,C);
// axis 0: out (B,out,W,C) from in (B,in_h,W,C).  planar (optional): float (B,C,oh,ow) = out / 255.
__global__ __launch_bounds__(256) void resize_pass_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                             float* __restrict__ planar, const int* __restrict__ bounds,
                                                             const int* __restrict__ coeffs, int ksize, int B, int in_h, int in_w,
                                                             int C, int out_size, int axis) {
    const int oh = axis == 0 ? out_size : in_h, ow = axis == 1 ? out_size : in_w;
    const int total = B * oh * ow;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int x = i % ow, y = (i / ow) % oh, b = i / (ow * oh);
        const int o = axis == 1 ? x : y;
        const int first = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = coeffs + (size_t)o * ksize;
        int acc[4] = {1 << (RS_PRECISION_BITS - 1), 1 << (RS_PRECISION_BITS - 1), 1 << (RS_PRECISION_BITS - 1),
                      1 << (RS_PRECISION_BITS - 1)};
        const size_t step = axis == 1 ? (size_t)C : (size_t)in_w * C;
        const unsigned char* p = src + (((size_t)b * in_h + (axis == 1 ? y : first)) * in_w + (axis == 1 ? first : x)) * C;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < C) acc[c] += (int)p[c] * w;
            p += step;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c >= C) continue;
            const int v = min(max(acc[c] >> RS_PRECISION_BITS, 0), 255);
            dst[(((size_t)b * oh + y) * ow + x) * C + c] = (unsigned char)v;
            if (planar) planar[(((size_t)b * C + c) * oh + y) * ow + x] = (float)v / 255.f;   // ToTensor
        }
    }
}

// ToTensor alone (level 0 when the raw image already has the model's size)
__global__ __launch_bounds__(256) void u8_to_planar_kernel(const unsigned char* __restrict__ src, float* __restrict__ planar, int B,
                                                           int H, int W, int C) {
    const int total = B * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int x = i % W, y = (i / W) % H, b = i / (W * H);
        for (int c = 0; c < C; ++c)
            planar[(((size_t)b * C + c) * H + y) * W + x] = (float)src[(size_t)i * C + c] / 255.f;
    }
}

// ------------------------------------------------------------------------------------------------
// Colour jitter (datasets/utils.py:236-259: torchvision adjust_{brightness,contrast,saturation,hue} on PIL
// images in a random order; restated with Pillow's exact C arithmetic in oracle/jitter.py).  uint8 RGB
// interleaved; every op is one elementwise kernel, contrast additionally needs the image's mean luminance
// (exact integer sum by 64-bit atomics: order-independent, hence deterministic).
__device__ __forceinline__ int lum_u8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Image.blend(other, img, alpha): float32 multiply and add, NOT fused (Pillow's C is not contracted);
// alpha in [0,1] truncates, otherwise clips.
__device__ __forceinline__ unsigned char blend_u8(int other, int v, float alpha, bool interp) {
    const float m = alpha * (float)(v - other);    // separate roundings (contraction is off in this file)
    const float t = (float)other + m;
    if (interp) return (unsigned char)t;
    return t <= 0.f ? (unsigned char)0 : (t >= 255.f ? (unsigned char)255 : (unsigned char)t);
}

__global__ __launch_bounds__(256) void jitter_lsum_kernel(const unsigned char* __restrict__ img, unsigned long long* __restrict__ lsum,
                                                          int HW) {
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    unsigned long long s = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const unsigned char* p = img + ((size_t)b * HW + i) * 3;
        s += (unsigned long long)lum_u8(p[0], p[1], p[2]);
    }
    // block reduction through LDS (integers: any order gives the same sum)
    for (int off = 32; off >= 1; off >>= 1) {
        const float lo = wave_shfl_xor(__uint_as_float((unsigned)(s & 0xffffffffu)), off);
        const float hi = wave_shfl_xor(__uint_as_float((unsigned)(s >> 32)), off);
        s += ((unsigned long long)__float_as_uint(hi) << 32) | (unsigned long long)__float_as_uint(lo);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&lsum[b], red[0] + red[1] + red[2] + red[3]);
}

// mode 0: brightness (other = 0), 1: contrast (other = int(mean L + 0.5)), 2: saturation (other = L of the pixel)
__global__ __launch_bounds__(256) void jitter_blend_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           const unsigned long long* __restrict__ lsum, int B, int HW, int mode,
                                                           float alpha) {
    const bool interp = alpha >= 0.f && alpha <= 1.f;
    const int total = B * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned char* p = src + (size_t)i * 3;
        const int r = p[0], g = p[1], b = p[2];
        int other = 0;
        if (mode == 1) other = (int)((double)lsum[i / HW] / (double)HW + 0.5);    // ImageStat mean, Python float
        else if (mode == 2) other = lum_u8(r, g, b);
        unsigned char* o = dst + (size_t)i * 3;
        o[0] = blend_u8(other, r, alpha, interp);
        o[1] = blend_u8(other, g, alpha, interp);
        o[2] = blend_u8(other, b, alpha, interp);
    }
}

// Pillow's rgb2hsv_row -> h += shift (uint8 wrap) -> hsv2rgb, with its float / double mix
__global__ __launch_bounds__(256) void jitter_hue_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                         int total, int shift) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned char* p = src + (size_t)i * 3;
        const int r = p[0], g = p[1], b = p[2];
        const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
        int uh = 0, us = 0;
        const int uv = maxc;
        if (minc != maxc) {
            const float cr = (float)(maxc - minc);
            const float s = cr / (float)maxc;
            const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
            float h;
            if (r == maxc) h = bc - gc;
            else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
            else h = (float)(4.0 + (double)gc - (double)rc);
            h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
            uh = min(max((int)((double)h * 255.0), 0), 255);
            us = min(max((int)((double)s * 255.0), 0), 255);
        }
        uh = (uh + shift) & 255;
        unsigned char* o = dst + (size_t)i * 3;
        if (us == 0) {
            o[0] = o[1] = o[2] = (unsigned char)uv;
            continue;
        }
        const double h6 = (double)(float)uh * 6.0 / 255.0;
        const int ii = (int)floor(h6);
        const double f = (double)(float)(h6 - (double)(float)ii);
        const double fs = (double)(float)((double)(float)us / 255.0);
        const double v = (double)(float)uv;
        const int pp = min(max((int)round(v * (1.0 - fs)), 0), 255);
        const int qq = min(max((int)round(v * (1.0 - fs * f)), 0), 255);
        const int tt = min(max((int)round(v * (1.0 - fs * (1.0 - f))), 0), 255);
        int R, G, Bc;
        switch (ii % 6) {
            case 0: R = uv; G = tt; Bc = pp; break;
            case 1: R = qq; G = uv; Bc = pp; break;
            case 2: R = pp; G = uv; Bc = tt; break;
            case 3: R = pp; G = qq; Bc = uv; break;
            case 4: R = tt; G = pp; Bc = uv; break;
            default: R = uv; G = pp; Bc = qq; break;
        }
        o[0] = (unsigned char)R; o[1] = (unsigned char)G; o[2] = (unsigned char)Bc;
    }
}

static double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}

static double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

}  // namespace clslam

using namespace clslam;

// Taps per output index of a LANCZOS resize in_size -> out_size (the row length of the coefficient table).
extern "C" int clslam_lanczos_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(3.0 * filterscale) * 2 + 1;
}

// HOST function: bounds[out_size][2] = (first tap, tap count), coeffs[out_size][ksize] = 22-bit fixed-point
// weights, computed in double with libm like Pillow's precompute_coeffs + normalize_coeffs_8bpc.
extern "C" int clslam_lanczos_plan(int in_size, int out_size, int* bounds, int* coeffs) {
    CLSLAM_REQUIRE(in_size > 0 && out_size > 0 && bounds && coeffs, "lanczos_plan: bad args");
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = lanczos_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = xmax; x < ksize; ++x) k[x] = 0.0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
        for (int x = 0; x < ksize; ++x) {
            const double f = k[x] * (1 << RS_PRECISION_BITS);
            coeffs[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + f) : (int)(0.5 + f);
        }
    }
    return CLSLAM_OK;
}

// One pass of the separable resize on interleaved uint8 images (device pointers; bounds/coeffs = the plan
// uploaded by the caller).  axis 1 = horizontal (width in_w -> out_size), axis 0 = vertical (height in_h ->
// out_size).  planar (optional): additionally writes ToTensor(out) as float (B,C,oh,ow).
extern "C" int clslam_resize_pass_u8(const unsigned char* src, unsigned char* dst, float* planar, const int* bounds,
                                     const int* coeffs, int ksize, int batch, int in_h, int in_w, int ch, int out_size, int axis,
                                     void* stream) {
    if (batch == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(src && dst && bounds && coeffs && ksize > 0 && ch >= 1 && ch <= 4 && (axis == 0 || axis == 1) && out_size > 0,
                   "resize_pass_u8: bad args");
    const size_t total = (size_t)batch * (axis == 0 ? out_size : in_h) * (axis == 1 ? out_size : in_w);
    CLSLAM_REQUIRE(total < ((size_t)1 << 31), "resize_pass_u8: image too large for 32-bit indexing");
    if (!total) return CLSLAM_OK;
    const unsigned blocks = (unsigned)std::min<size_t>(8192, (total + 255) / 256);
    hipLaunchKernelGGL(resize_pass_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, planar, bounds, coeffs,
                       ksize, batch, in_h, in_w, ch, out_size, axis);
    return check_launch("resize_pass_u8");
}

extern "C" int clslam_u8_to_planar_f32(const unsigned char* src, float* planar, int batch, int h, int w, int ch, void* stream) {
    if (batch == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(src && planar && ch >= 1, "u8_to_planar_f32: bad args");
    const size_t total = (size_t)batch * h * w;
    CLSLAM_REQUIRE(total < ((size_t)1 << 31), "u8_to_planar_f32: image too large for 32-bit indexing");
    if (!total) return CLSLAM_OK;
    const unsigned blocks = (unsigned)std::min<size_t>(8192, (total + 255) / 256);
    hipLaunchKernelGGL(u8_to_planar_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, planar, batch, h, w, ch);
    return check_launch("u8_to_planar_f32");
}

// Colour jitter of interleaved uint8 RGB images (B,h,w,3).  order[n_ops]: op ids in application order
// (0 brightness, 1 contrast, 2 saturation, 3 hue), factors[4] (double, as Python draws them) indexed by op id
// (HOST arrays).  scratch: a
// second image buffer of the same size (ops ping-pong, the result ends in dst); lsum: batch uint64 of
// device scratch for the contrast mean.  src is not modified.
extern "C" int clslam_color_jitter_u8(const unsigned char* src, unsigned char* dst, unsigned char* scratch,
                                      unsigned long long* lsum, int batch, int h, int w, const int* order, int n_ops,
                                      const double* factors, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CLSLAM_REQUIRE(order && factors && n_ops >= 0 && n_ops <= 8, "color_jitter_u8: bad args");
    for (int k = 0; k < n_ops; ++k) CLSLAM_REQUIRE(order[k] >= 0 && order[k] <= 3, "color_jitter_u8: unknown op %d", order[k]);
    CLSLAM_REQUIRE(n_ops == 0 || (factors[3] >= -0.5 && factors[3] <= 0.5), "color_jitter_u8: hue_factor outside [-0.5, 0.5]");
    if (batch == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(src && dst && scratch && lsum, "color_jitter_u8: null pointer");
    const size_t total = (size_t)batch * h * w;
    CLSLAM_REQUIRE(total < ((size_t)1 << 30), "color_jitter_u8: image too large for 32-bit indexing");
    if (!total) return CLSLAM_OK;
    const unsigned blocks = (unsigned)std::min<size_t>(4096, (total + 255) / 256);
    if (n_ops == 0) {
        hipMemcpyAsync(dst, src, total * 3, hipMemcpyDeviceToDevice, stream);
        return check_launch("color_jitter_u8");
    }
    const unsigned char* cur = src;
    for (int k = 0; k < n_ops; ++k) {
        unsigned char* out = ((n_ops - 1 - k) % 2 == 0) ? dst : scratch;     // the last op writes dst
        const int op = order[k];
        if (op == 3) {
            const int shift = ((int)(factors[3] * 255.0) % 256 + 256) % 256;   // np.uint8(hue_factor * 255): truncate, wrap
            hipLaunchKernelGGL(jitter_hue_kernel, dim3(blocks), dim3(256), 0, stream, cur, out, (int)total, shift);
        } else {
            if (op == 1) {
                hipMemsetAsync(lsum, 0, sizeof(unsigned long long) * batch, stream);
                hipLaunchKernelGGL(jitter_lsum_kernel, dim3(std::min(64, cdiv(h * w, 256)), batch), dim3(256), 0, stream, cur, lsum,
                                   h * w);
            }
            hipLaunchKernelGGL(jitter_blend_kernel, dim3(blocks), dim3(256), 0, stream, cur, out, lsum, batch, h * w, op,
                               (float)factors[op]);   // Image.blend takes a C float
        }
        cur = out;
    }
    return check_launch("color_jitter_u8");
}
