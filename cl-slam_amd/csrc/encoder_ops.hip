// K1/K2: ResNet stem for the frozen depth / pose encoders.
//   stem   : (x - 0.45)/0.225 (reference networks/resnet_encoder.py:117) -> conv7x7 s2 p3, no bias
//            (:118, torchvision ResNet.conv1 / ResNetMultiImageInput :25-30) -> eval BatchNorm
//            (scale/shift) -> ReLU (:119-120), reading the PLANAR NCHW images of the reference's
//            sample dict directly (for the pose net the two frames are two pointers, the
//            torch.cat of dpp.py:951-955 never materialises) and writing NHWC.
//   maxpool: 3x3 s2 p1 (:121).
// (The conv weight arrives pre-packed: clslam_stem_pack_weight, once per load.)
// The stem is an implicit GEMM on v_mfma_f32_32x32x2_f32: M = 8x16 output pixels per block,
// N = 64 output channels, K = 3*49 (+1 zero) per pass of three input channels; per pass the 3x21x37
// input patch (normalised, zero outside the image = zero padding of the normalised tensor) and the
// 64x147 weight slab are staged in LDS (47 KB -> 3 blocks per CU).
#include "common.h"

namespace clslam {

constexpr int ST_TH = 8, ST_TW = 16;                  // output tile
constexpr int ST_PH = 2 * ST_TH + 5, ST_PW = 2 * ST_TW + 5;  // 21 x 37 input patch per channel
constexpr int ST_CG = 3;                              // input channels staged per pass
constexpr int ST_K = ST_CG * 49;                      // 147 reduction elements per pass (+1 zero)
constexpr int ST_LDW = 149;                           // weight row stride (odd -> conflict-free column reads)

// LDS patch offset of reduction element k = (c, ky, kx) of a pass
__device__ __forceinline__ constexpr int st_koff(int k) {
    return (k / 49) * (ST_PH * ST_PW) + ((k % 49) / 7) * ST_PW + (k % 7);
}

__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ img_a, const float* __restrict__ img_b,
                                                        const float* __restrict__ w, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ out,
                                                        int B, int H, int W, int Cin, int Ho, int Wo, int tiles_x,
                                                        int tiles_y) {
    __shared__ float patch[ST_CG * ST_PH * ST_PW];
    __shared__ __attribute__((aligned(16))) float Ws[64 * ST_LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid;
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int i = lane & 31, g = lane >> 5;
    const int py = wave * 2 + (i >> 4), px = i & 15;      // this lane's output pixel inside the tile
    const int pbase = (2 * py) * ST_PW + 2 * px;

    // one pass per group of 3 input channels (depth net: 1 pass, pose net: 2): stage the 3 x 21 x 37
    // normalised patch and the 64 x 147 weight slab, then 74 MFMA k-steps without further barriers
    constexpr int P_N = ST_CG * ST_PH * ST_PW, P_IT = (P_N + 255) / 256;      // 2331 patch elements, 10 per thread
    constexpr int W_N4 = 64 * ST_LDW / 4, W_IT = (W_N4 + 255) / 256;         // 2384 float4 of weights, 10 per thread
    static_assert(64 * ST_LDW % 4 == 0, "weight slab is copied as float4");
    for (int c0 = 0; c0 < Cin; c0 += ST_CG) {
        // All global loads of the pass are issued back to back into registers (20 independent loads in
        // flight per thread), then written to LDS: a load->store loop pays the memory latency ~10x per block.
        float pv[P_IT];
        bool pok[P_IT];
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int e = tid + it * 256;
            const int ee = e < P_N ? e : 0;
            const int c = ee / (ST_PH * ST_PW), rem = ee - c * (ST_PH * ST_PW);
            const int r = rem / ST_PW, q = rem - r * ST_PW;
            const int iy = iy0 + r, ix = ix0 + q;
            const int cc = c0 + c;
            const float* img = (cc < 3) ? img_a + ((size_t)b * 3 + cc) * H * W : img_b + ((size_t)b * 3 + (cc - 3)) * H * W;
            pok[it] = iy >= 0 && iy < H && ix >= 0 && ix < W;
            pv[it] = img[pok[it] ? (size_t)iy * W + ix : 0];
        }
        // weight slab of this pass, pre-packed by the host in the LDS image [64][ST_LDW] (zero padded)
        const f32x4* wp4 = reinterpret_cast<const f32x4*>(w + (size_t)(c0 / ST_CG) * 64 * ST_LDW);
        f32x4 wv[W_IT];   // native vector type: the HIP float4 struct array ended up in scratch here
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int e = tid + it * 256;
            wv[it] = wp4[e < W_N4 ? e : 0];
        }
        __syncthreads();   // previous pass's MFMAs are done with LDS
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int e = tid + it * 256;
            if (e < P_N) patch[e] = pok[it] ? (pv[it] - 0.45f) / 0.225f : 0.f;
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int e = tid + it * 256;
            if (e < W_N4) reinterpret_cast<f32x4*>(Ws)[e] = wv[it];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < (ST_K + 1) / 2; ++s) {
            // k = 2s + g; k = 147 is the zero pad column (any in-range patch element will do)
            constexpr int dummy = 0;
            const int off0 = st_koff(2 * s);
            const int off1 = (2 * s + 1 < ST_K) ? st_koff(2 * s + 1) : dummy;
            const float a = patch[pbase + (g ? off1 : off0)];
            const float b0 = Ws[i * ST_LDW + 2 * s + g];
            const float b1 = Ws[(32 + i) * ST_LDW + 2 * s + g];
            acc[0] = mfma_32x32x2(a, b0, acc[0]);
            acc[1] = mfma_32x32x2(a, b1, acc[1]);
        }
    }
    // Epilogue in two straight-line phases: every value first (BN scale / shift of both channel halves loaded up front), then
    // the 32 stores back to back.  With `load scale/shift -> (value, conditional store) x 16` per half hipcc put an
    // `s_waitcnt vmcnt(0)` in front of EVERY store: 32 serial memory round trips per workgroup (round 5).
    float sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { sc[j] = scale[j * 32 + i]; sh[j] = shift[j * 32 + i]; }
    // (the four loads are consumed HERE: hipcc sinks the value computation into the conditional store blocks otherwise and
    // guards each of them with `s_waitcnt vmcnt(2)` -- which, stores counting in vmcnt, keeps at most two stores in flight)
    consume_now(sc[0]); consume_now(sh[0]); consume_now(sc[1]); consume_now(sh[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[j][r] * sc[j] + sh[j];
            acc[j][r] = v > 0.f ? v : 0.f;
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;  // pixel index inside the wave's 32
            const int oy = oy0 + wave * 2 + (row >> 4), ox = ox0 + (row & 15);
            if (oy < Ho && ox < Wo) out[(((size_t)b * Ho + oy) * Wo + ox) * 64 + n] = acc[j][r];
        }
    }
}

__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C,
                                    int Ho, int Wo) {
    const int C4 = C / 4;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int ox = (int)((idx / C4) % Wo);
        const int oy = (int)((idx / ((size_t)C4 * Wo)) % Ho);
        const int b = (int)(idx / ((size_t)C4 * Wo * Ho));
        // Nine UNCONDITIONAL loads from clamped positions, all in flight together: a tap outside the image repeats one inside
        // the window (the centre (2oy, 2ox) always is), which leaves the maximum unchanged.  With `continue` per tap every
        // load sat in its own block behind an `s_waitcnt vmcnt(0)`: nine serial memory round trips per output (round 5).
        float4 v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = min(max(2 * oy - 1 + t / 3, 0), H - 1), ix = min(max(2 * ox - 1 + t % 3, 0), W - 1);
            v[t] = *reinterpret_cast<const float4*>(in + (((size_t)b * H + iy) * W + ix) * C + c4 * 4);
        }
        float4 m = v[4];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            m.x = fmaxf(m.x, v[t].x); m.y = fmaxf(m.y, v[t].y); m.z = fmaxf(m.z, v[t].z); m.w = fmaxf(m.w, v[t].w);
        }
        *reinterpret_cast<float4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * C + c4 * 4) = m;
    }
}

}  // namespace clslam

using namespace clslam;

__global__ void stem_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cin) {
    // packed[pass][co][k] = w[co][pass*3 + k/49][k%49]  (k < 147), 0 for the pad columns
    const int total = (Cin / ST_CG) * 64 * ST_LDW;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int k = e % ST_LDW, co = (e / ST_LDW) % 64, ps = e / (ST_LDW * 64);
        packed[e] = (k < ST_K) ? w[((size_t)co * Cin + ps * ST_CG) * 49 + k] : 0.f;
    }
}

extern "C" int clslam_stem_packed_size(int num_images) { return num_images * 64 * ST_LDW; }

// weight: OIHW (64, 3*num_images, 7, 7) as in the checkpoint -> packed slabs for clslam_stem_conv
extern "C" int clslam_stem_pack_weight(const float* weight, float* packed, int num_images, void* stream) {
    CLSLAM_REQUIRE(weight && packed && (num_images == 1 || num_images == 2), "stem_pack_weight: bad args");
    hipLaunchKernelGGL(stem_pack_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, weight, packed, 3 * num_images);
    return check_launch("stem_pack_weight");
}

extern "C" int clslam_stem_conv(const float* img_a, const float* img_b, const float* weight, const float* scale,
                                const float* shift, float* out, int batch, int h, int w, int num_images, void* stream) {
    CLSLAM_REQUIRE(img_a && weight && scale && shift && out, "stem_conv: null pointer");
    CLSLAM_REQUIRE(num_images == 1 || (num_images == 2 && img_b), "stem_conv: num_images must be 1 or 2");
    const int Ho = (h + 6 - 7) / 2 + 1, Wo = (w + 6 - 7) / 2 + 1;
    const int tx = cdiv(Wo, ST_TW), ty = cdiv(Ho, ST_TH);
    if (batch == 0) return CLSLAM_OK;
    hipLaunchKernelGGL(stem_conv_kernel, dim3(tx * ty * batch), dim3(256), 0, (hipStream_t)stream, img_a, img_b, weight,
                       scale, shift, out, batch, h, w, 3 * num_images, Ho, Wo, tx, ty);
    return check_launch("stem_conv");
}

extern "C" int clslam_maxpool3x3s2(const float* in, float* out, int batch, int h, int w, int ch, void* stream) {
    if (batch == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(in && out && ch % 4 == 0, "maxpool: bad args");
    const int Ho = (h + 2 - 3) / 2 + 1, Wo = (w + 2 - 3) / 2 + 1;
    const size_t total = (size_t)batch * Ho * Wo * (ch / 4);
    if (total == 0) return CLSLAM_OK;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, in, out, batch, h, w, ch, Ho, Wo);
    return check_launch("maxpool3x3s2");
}
