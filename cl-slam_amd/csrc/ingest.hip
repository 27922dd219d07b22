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

// =====================================================================================================================
// The replay buffer's colour jitter: torchvision's TENSOR code path on float images in [0, 1] (slam/replay_buffer.py:264-265,
// 281-283 after ToTensor; slam/slam.py:98 do_augmentation=True) -- functional_tensor.py of torchvision 0.11.1, operation by
// operation (oracle/jitter_tensor.py is the checker): _blend = (ratio * a + (1 - ratio) * b).clamp(0, 1) in float32, gray =
// 0.2989 r + 0.587 g + 0.114 b, contrast blends with the image's mean gray, hue = _rgb2hsv -> (h + f) % 1 -> _hsv2rgb.
// Contraction is off for this file: every product and sum rounds where torch's separate kernels round.
// One image = one (3, h, w) planar float plane set with ITS OWN op order and factors (a replayed sample draws one jitter for its
// three frames and four scales; the contrast mean is per image).  Two launches for any number of images: pass 1 applies the ops
// in front of `contrast` and leaves per-block sums of the gray value, pass 2 forms the mean (fixed order) and applies the chain.
namespace clslam {

struct JitterParams {        // per image, device memory: order[k] in {0 brightness, 1 contrast, 2 saturation, 3 hue, -1 end}
    int order[4];
    float f[4];              // factor as a C float, indexed by op id
    float omf[4];            // (float)(1.0 - factor): torch forms 1.0 - ratio in double before the multiply
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return (0.2989f * r + 0.587f * g) + 0.114f * b; }

__device__ __forceinline__ void jitter_hue(float& r, float& g, float& b, float hf) {
    // _rgb2hsv
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const bool eqc = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eqc ? 1.f : maxc);
    const float crd = eqc ? 1.f : cr;
    const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
    const float hr = (maxc == r) ? (bc - gc) : 0.f;
    const float hg = ((maxc == g) && (maxc != r)) ? ((2.0f + rc) - bc) : 0.f;
    const float hb = ((maxc != g) && (maxc != r)) ? ((4.0f + gc) - rc) : 0.f;
    float h = (hr + hg) + hb;
    h = fmodf(h / 6.0f + 1.0f, 1.0f);
    // (h + hue_factor) % 1.0: torch.remainder = fmod, then + 1 when the sign differs from the divisor's
    float hh = fmodf(h + hf, 1.0f);
    if (hh != 0.f && hh < 0.f) hh += 1.0f;
    // _hsv2rgb
    const float v = maxc;
    const float h6 = hh * 6.0f;
    const float fl = floorf(h6);
    const float f = h6 - fl;
    int i = (int)fl;
    const float p = clamp01(v * (1.0f - s));
    const float q = clamp01(v * (1.0f - f * s));
    const float t = clamp01(v * (1.0f - (s * (1.0f - f))));
    i = ((i % 6) + 6) % 6;
    r = i == 0 ? v : i == 1 ? q : i == 2 ? p : i == 3 ? p : i == 4 ? t : v;
    g = i == 0 ? t : i == 1 ? v : i == 2 ? v : i == 3 ? q : i == 4 ? p : p;
    b = i == 0 ? p : i == 1 ? p : i == 2 ? t : i == 3 ? v : i == 4 ? v : q;
}

// ops [k0, k1) of the chain on one pixel; `mean` is used by the contrast op only
__device__ __forceinline__ void jitter_ops(const JitterParams& P, int k0, int k1, float& r, float& g, float& b, float mean) {
    for (int k = k0; k < k1; ++k) {
        const int op = P.order[k];
        if (op < 0) break;
        const float f = P.f[op], omf = P.omf[op];
        if (op == 0) {
            r = clamp01(f * r + omf * 0.f); g = clamp01(f * g + omf * 0.f); b = clamp01(f * b + omf * 0.f);
        } else if (op == 1) {
            const float m = omf * mean;
            r = clamp01(f * r + m); g = clamp01(f * g + m); b = clamp01(f * b + m);
        } else if (op == 2) {
            const float m = omf * gray_of(r, g, b);
            r = clamp01(f * r + m); g = clamp01(f * g + m); b = clamp01(f * b + m);
        } else {
            jitter_hue(r, g, b, f);
        }
    }
}

__device__ __forceinline__ int jitter_contrast_pos(const JitterParams& P) {
    for (int k = 0; k < 4; ++k) {
        if (P.order[k] < 0) break;
        if (P.order[k] == 1) return k;
    }
    return -1;
}

// grid (nblk, images).  partial[img][blk] = sum of the gray value in front of the contrast op over the block's pixels
__global__ __launch_bounds__(256) void jitter_f32_sum_kernel(const float* __restrict__ src, const JitterParams* __restrict__ params,
                                                             float* __restrict__ partial, int hw) {
    __shared__ float red[4];
    const int img = blockIdx.y;
    const JitterParams P = params[img];
    const int kc = jitter_contrast_pos(P);
    if (kc < 0) return;
    const float* base = src + (size_t)img * 3 * hw;
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        float r = base[i], g = base[hw + i], b = base[2 * hw + i];
        jitter_ops(P, 0, kc, r, g, b, 0.f);
        acc += gray_of(r, g, b);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)img * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void jitter_f32_apply_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               const JitterParams* __restrict__ params, const float* __restrict__ partial,
                                                               int nblk_sum, int hw) {
    __shared__ float s_mean;
    const int img = blockIdx.y;
    const JitterParams P = params[img];
    if (threadIdx.x == 0) {
        float m = 0.f;
        if (jitter_contrast_pos(P) >= 0) {
            for (int k = 0; k < nblk_sum; ++k) m += partial[(size_t)img * nblk_sum + k];     // fixed order
            m = m / (float)hw;
        }
        s_mean = m;
    }
    __syncthreads();
    const float mean = s_mean;
    const float* base = src + (size_t)img * 3 * hw;
    float* out = dst + (size_t)img * 3 * hw;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        float r = base[i], g = base[hw + i], b = base[2 * hw + i];
        jitter_ops(P, 0, 4, r, g, b, mean);
        out[i] = r; out[hw + i] = g; out[2 * hw + i] = b;
    }
}

}  // namespace clslam

extern "C" int clslam_color_jitter_f32_blocks(int h, int w) { return std::max(1, std::min(64, clslam::cdiv(h * w, 1024))); }

extern "C" int clslam_color_jitter_f32(const float* src, float* dst, const void* params, float* partial, int n_images, int h, int w,
                                       void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_images == 0 || h * w == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(src && dst && params && partial, "color_jitter_f32: null pointer");
    CLSLAM_REQUIRE(n_images > 0 && n_images <= 65535 && h > 0 && w > 0, "color_jitter_f32: bad geometry");
    const int nblk = clslam_color_jitter_f32_blocks(h, w);
    hipLaunchKernelGGL(clslam::jitter_f32_sum_kernel, dim3(nblk, n_images), dim3(256), 0, stream, src, (const clslam::JitterParams*)params,
                       partial, h * w);
    const int nb2 = std::max(1, std::min(256, clslam::cdiv(h * w, 512)));
    hipLaunchKernelGGL(clslam::jitter_f32_apply_kernel, dim3(nb2, n_images), dim3(256), 0, stream, src, dst,
                       (const clslam::JitterParams*)params, partial, nblk, h * w);
    return clslam::check_launch("color_jitter_f32");
}
