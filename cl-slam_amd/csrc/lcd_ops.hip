// K20: the non-GEMM ops of the loop-closure feature encoder (MobileNetV3-small forward,
// reference loop_closure_detection/encoder.py:13-33; B = 1 per keyframe, ~0.14 GMAC at 192x640, HBM /
// latency bound).  The pointwise 1x1 convolutions go through clslam_conv2d.
#include "common.h"

namespace clslam {

// (x - mean)/std (encoder.py:14,29) -> conv3x3 s2 p1 3->16 -> BN -> hardswish; one thread per output pixel.
__global__ __launch_bounds__(256) void mbv3_stem_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out, int B, int H, int W, int Ho, int Wo) {
    __shared__ float ws[27 * 16];   // [c*9+tap][cout]
    __shared__ float sc[16], sh[16];
    for (int e = threadIdx.x; e < 27 * 16; e += 256) {
        const int co = e % 16, k = e / 16;      // k = c*9 + ky*3 + kx  (OIHW: w[co][c][ky][kx])
        ws[e] = w[co * 27 + k];
    }
    if (threadIdx.x < 16) { sc[threadIdx.x] = scale[threadIdx.x]; sh[threadIdx.x] = shift[threadIdx.x]; }
    __syncthreads();
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const size_t total = (size_t)B * Ho * Wo;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho), b = (int)(idx / ((size_t)Wo * Ho));
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float* pl = img + ((size_t)b * 3 + c) * H * W;
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = 2 * oy - 1 + ky;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = 2 * ox - 1 + kx;
                    if (ix < 0 || ix >= W) continue;
                    const float v = (pl[(size_t)iy * W + ix] - mean[c]) / stdv[c];
                    const float* wr = ws + (c * 9 + ky * 3 + kx) * 16;
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[k] = fmaf(v, wr[k], acc[k]);
                }
            }
        }
        float* o = out + idx * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = apply_act(acc[k] * sc[k] + sh[k], CLSLAM_ACT_HSWISH);
    }
}

__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ out, int B, int H, int W, int C, int Ho, int Wo,
                                                     int K, int stride, int act) {
    const int C4 = C / 4, pad = K / 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int ox = (int)((idx / C4) % Wo);
        const int oy = (int)((idx / ((size_t)C4 * Wo)) % Ho);
        const int b = (int)(idx / ((size_t)C4 * Wo * Ho));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + iy) * W + ix) * C + c4 * 4);
                const float4 k = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + c4 * 4);
                acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y);
                acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
            }
        }
        const float4 s = *reinterpret_cast<const float4*>(scale + c4 * 4);
        const float4 t = *reinterpret_cast<const float4*>(shift + c4 * 4);
        float4 r;
        r.x = apply_act(acc.x * s.x + t.x, act); r.y = apply_act(acc.y * s.y + t.y, act);
        r.z = apply_act(acc.z * s.z + t.z, act); r.w = apply_act(acc.w * s.w + t.w, act);
        *reinterpret_cast<float4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4) = r;
    }
}

// Stage 1, grid (ceil(C/64), B, chunks): 16 channel-quads x 16 pixel lanes per block over one pixel chunk;
// writes the chunk sum to partial[b][chunk][C] (or, chunks == 1, the mean straight to out).  One block per
// (sample, 64 channels) walking all H*W pixels was a 40-us chain of dependent-latency loads at B = 1.
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             float* __restrict__ partial, int HW, int C, int chunk_px) {
    __shared__ float4 red[256];
    const int b = blockIdx.y, ch = blockIdx.z, nch = gridDim.z;
    const int q = threadIdx.x % 16, pl = threadIdx.x / 16;
    const int c4 = blockIdx.x * 16 + q;
    const int p0 = ch * chunk_px, p1 = min(HW, p0 + chunk_px);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 * 4 < C)
        for (int p = p0 + pl; p < p1; p += 16) {
            const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HW + p) * C + c4 * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && c4 * 4 < C) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < 16; ++k) { const float4 v = red[k * 16 + q]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        if (nch == 1) {
            const float inv = 1.f / (float)HW;
            t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
            *reinterpret_cast<float4*>(out + (size_t)b * C + c4 * 4) = t;
        } else {
            *reinterpret_cast<float4*>(partial + ((size_t)b * nch + ch) * C + c4 * 4) = t;
        }
    }
}

// Stage 2: out[b][c] = (sum over chunks, in order) / HW
__global__ __launch_bounds__(256) void avgpool_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int B,
                                                             int C, int nch, int HW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int k = 0; k < nch; ++k) s += partial[((size_t)b * nch + k) * C + c];
    out[i] = s / (float)HW;
}

// grid (ceil(C/64), B).  Every block recomputes hidden = relu(w1*pool+b1) (S <= 256; 16 lanes per hidden unit,
// shuffle-reduced) and then produces 64 gate channels, gate = hardsigmoid(w2*hidden+b2), 4 lanes per channel.
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ pool, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ gate, int C, int S) {
    __shared__ float hid[256];
    const int b = blockIdx.y;
    const float* pv = pool + (size_t)b * C;
    const int sl = threadIdx.x >> 4, cl = threadIdx.x & 15;
    for (int s0 = 0; s0 < S; s0 += 16) {          // uniform trip count: the shuffles need every lane
        const int s = s0 + sl;
        float a = 0.f;
        if (s < S)
            for (int c = cl; c < C; c += 16) a = fmaf(w1[(size_t)s * C + c], pv[c], a);
        a += wave_shfl_xor(a, 1); a += wave_shfl_xor(a, 2); a += wave_shfl_xor(a, 4); a += wave_shfl_xor(a, 8);
        if (s < S && cl == 0) { a += b1[s]; hid[s] = a > 0.f ? a : 0.f; }
    }
    __syncthreads();
    const int c = blockIdx.x * 64 + (threadIdx.x >> 2), kl = threadIdx.x & 3;
    float a = 0.f;
    if (c < C)
        for (int s = kl; s < S; s += 4) a = fmaf(w2[(size_t)c * S + s], hid[s], a);
    a += wave_shfl_xor(a, 1); a += wave_shfl_xor(a, 2);
    if (c < C && kl == 0) gate[(size_t)b * C + c] = apply_act(a + b2[c], CLSLAM_ACT_HSIGMOID);
}

__global__ __launch_bounds__(256) void channel_scale_kernel(float* __restrict__ x, const float* __restrict__ gate, int B,
                                                            int HW, int C) {
    const int C4 = C / 4;
    const size_t total = (size_t)B * HW * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int b = (int)(idx / ((size_t)C4 * HW));
        float4 v = reinterpret_cast<float4*>(x)[idx];
        const float4 g = *reinterpret_cast<const float4*>(gate + (size_t)b * C + c4 * 4);
        v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
        reinterpret_cast<float4*>(x)[idx] = v;
    }
}

}  // namespace clslam

using namespace clslam;

static unsigned lgrid(size_t total) { return (unsigned)std::max<size_t>(1, std::min<size_t>(4096, (total + 255) / 256)); }

extern "C" int clslam_mbv3_stem(const float* img, const float* weight, const float* scale, const float* shift, float* out,
                                int batch, int h, int w, void* stream) {
    CLSLAM_REQUIRE(img && weight && scale && shift && out, "mbv3_stem: null");
    const int Ho = (h + 2 - 3) / 2 + 1, Wo = (w + 2 - 3) / 2 + 1;
    const size_t total = (size_t)batch * Ho * Wo;
    if (!total) return CLSLAM_OK;
    hipLaunchKernelGGL(mbv3_stem_kernel, dim3(lgrid(total)), dim3(256), 0, (hipStream_t)stream, img, weight, scale, shift, out,
                       batch, h, w, Ho, Wo);
    return check_launch("mbv3_stem");
}

extern "C" int clslam_dwconv(const float* x, const float* weight, const float* scale, const float* shift, float* out, int batch,
                             int h, int w, int ch, int ksize, int stride, int act, void* stream) {
    CLSLAM_REQUIRE(x && weight && scale && shift && out && ch % 4 == 0, "dwconv: bad args");
    CLSLAM_REQUIRE((ksize == 3 || ksize == 5) && (stride == 1 || stride == 2), "dwconv: ksize 3|5, stride 1|2");
    const int pad = ksize / 2;
    const int Ho = (h + 2 * pad - ksize) / stride + 1, Wo = (w + 2 * pad - ksize) / stride + 1;
    const size_t total = (size_t)batch * Ho * Wo * (ch / 4);
    if (!total) return CLSLAM_OK;
    hipLaunchKernelGGL(dwconv_kernel, dim3(lgrid(total)), dim3(256), 0, (hipStream_t)stream, x, weight, scale, shift, out, batch,
                       h, w, ch, Ho, Wo, ksize, stride, act);
    return check_launch("dwconv");
}

// Pixel chunks clslam_global_avgpool splits H*W into when given a partial buffer (batch*chunks*ch floats).
extern "C" int clslam_avgpool_chunks(int hw) { return std::max(1, std::min(64, hw / 64)); }

extern "C" int clslam_global_avgpool(const float* x, float* out, float* partial, int batch, int hw, int ch, void* stream) {
    if (!batch) return CLSLAM_OK;
    CLSLAM_REQUIRE(x && out && ch % 4 == 0 && hw > 0, "global_avgpool: bad args");
    const int nch = partial ? clslam_avgpool_chunks(hw) : 1;
    hipLaunchKernelGGL(global_avgpool_kernel, dim3(cdiv(ch, 64), batch, nch), dim3(256), 0, (hipStream_t)stream, x, out, partial,
                       hw, ch, cdiv(hw, nch));
    if (nch > 1)
        hipLaunchKernelGGL(avgpool_finish_kernel, dim3(cdiv(batch * ch, 256)), dim3(256), 0, (hipStream_t)stream, partial, out,
                           batch, ch, nch, hw);
    return check_launch("global_avgpool");
}

extern "C" int clslam_se_gate(const float* pool, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                              int batch, int ch, int squeeze, void* stream) {
    CLSLAM_REQUIRE(pool && w1 && b1 && w2 && b2 && gate && squeeze <= 256, "se_gate: bad args");
    if (!batch) return CLSLAM_OK;
    hipLaunchKernelGGL(se_gate_kernel, dim3(cdiv(ch, 64), batch), dim3(256), 0, (hipStream_t)stream, pool, w1, b1, w2, b2, gate, ch,
                       squeeze);
    return check_launch("se_gate");
}

extern "C" int clslam_channel_scale(float* x, const float* gate, int batch, int hw, int ch, void* stream) {
    CLSLAM_REQUIRE(x && gate && ch % 4 == 0, "channel_scale: bad args");
    const size_t total = (size_t)batch * hw * (ch / 4);
    if (!total) return CLSLAM_OK;
    hipLaunchKernelGGL(channel_scale_kernel, dim3(lgrid(total)), dim3(256), 0, (hipStream_t)stream, x, gate, batch, hw, ch);
    return check_launch("channel_scale");
}
