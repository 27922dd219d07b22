// K6/K16 and K7-tail/K17-head: the two thin output heads.
//   dispconv : reflection-padded 3x3 conv C -> 1 + sigmoid (reference networks/depth_decoder.py:67-69,
//              layers.py:28-48) -- a 9*C dot product per pixel, HBM-bound, no MFMA (N = 1).
//   pose head: pose_2 1x1 conv 256 -> 12, spatial mean, x0.01 (reference networks/pose_decoder.py:44-54).
//              mean and the 1x1 conv commute (both linear), so the head is mean -> 12x256 matvec.
// plus their backward passes (autograd of the same lines).
#include "common.h"

namespace clslam {

// ------------------------------------------------------------------------------------------------
// disp[b,y,x] = sigmoid(bias + sum_{tap,c} x[b, refl(y+ky-1), refl(x+kx-1), c] * w[tap][c])
// A workgroup owns a TH x TW pixel tile: its (TH+2) x (TW+2) x C input patch is staged ONCE in LDS with
// contiguous 16-byte loads (reflection resolved while staging), so every input element crosses L2 about 1.4
// times instead of 9.  C/4 lanes then cooperate on one pixel (one float4 of channels each, 9 taps from LDS --
// the lanes of a wave read 1 KiB of consecutive LDS, conflict-free), a butterfly over those lanes finishes it.
template <int C, int TW, int TH>
__global__ __launch_bounds__(256) void dispconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ disp,
                                                           int B, int H, int W, int tiles_x, int tiles_y) {
    constexpr int C4 = C / 4;                   // lanes per pixel: 4, 8, 16, 32
    constexpr int PPI = 256 / C4;               // pixels per pass of the workgroup
    constexpr int PW = TW + 2, PH = TH + 2;
    static_assert((TW * TH) % PPI == 0 && PPI <= TW * TH, "tile must be a whole number of passes");
    __shared__ float4 tile[PH * PW * C4];
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
    const int x0 = tx * TW, y0 = ty * TH;
    const int cq = threadIdx.x % C4, pl = threadIdx.x / C4;
    float4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(w + t * C + cq * 4);
    const float bv = bias[0];
    for (int e = threadIdx.x; e < PH * PW * C4; e += 256) {
        const int c4 = e % C4, px = (e / C4) % PW, py = e / (C4 * PW);
        // rows / columns past the image only occur in partial tiles and feed no stored pixel: clamp them
        const int iy = reflect_idx(min(y0 + py - 1, H), H), ix = reflect_idx(min(x0 + px - 1, W), W);
        tile[e] = *reinterpret_cast<const float4*>(x + (((size_t)b * H + iy) * W + ix) * C + c4 * 4);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < TW * TH / PPI; ++it) {
        const int p = it * PPI + pl;
        const int lx = p % TW, ly = p / TW;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 v = tile[((ly + ky) * PW + lx + kx) * C4 + cq];
                const float4 k = wv[ky * 3 + kx];
                acc = fmaf(v.x, k.x, acc); acc = fmaf(v.y, k.y, acc);
                acc = fmaf(v.z, k.z, acc); acc = fmaf(v.w, k.w, acc);
            }
        }
#pragma unroll
        for (int m = C4 >> 1; m >= 1; m >>= 1) acc += wave_shfl_xor(acc, m);
        const int xx = x0 + lx, yy = y0 + ly;
        if (cq == 0 && xx < W && yy < H) disp[((size_t)b * H + yy) * W + xx] = 1.f / (1.f + expf(-(acc + bv)));
    }
}

template <int C, int TW, int TH>
static int launch_dispconv_fwd(const float* x, const float* w, const float* bias, float* disp, int batch, int h, int wd,
                               hipStream_t stream) {
    const int tiles_x = cdiv(wd, TW), tiles_y = cdiv(h, TH);
    hipLaunchKernelGGL((dispconv_fwd_kernel<C, TW, TH>), dim3(tiles_x * tiles_y * batch), dim3(256), 0, stream, x, w, bias,
                       disp, batch, h, wd, tiles_x, tiles_y);
    return check_launch("dispconv_fwd");
}

// dxp[b,Py,Px,c] (+)= sum_{ky,kx} dz[b,Py-2+ky,Px-2+kx] * w[(2-ky)*3+(2-kx)][c]   (padded domain)
__global__ __launch_bounds__(256) void dispconv_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                float* __restrict__ dxp, int B, int H, int W, int C,
                                                                int accumulate) {
    __shared__ __attribute__((aligned(16))) float ws[9 * 256];
    for (int e = threadIdx.x; e < 9 * C; e += 256) ws[e] = w[e];
    __syncthreads();
    const int C4 = C / 4, Hp = H + 2, Wp = W + 2;
    const size_t total = (size_t)B * Hp * Wp * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        const int Px = (int)((idx / C4) % Wp);
        const int Py = (int)((idx / ((size_t)C4 * Wp)) % Hp);
        const int b = (int)(idx / ((size_t)C4 * Wp * Hp));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = Py - 2 + ky;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = Px - 2 + kx;
                if (xx < 0 || xx >= W) continue;
                const float g = dz[((size_t)b * H + yy) * W + xx];
                const float4 k = *reinterpret_cast<const float4*>(ws + ((2 - ky) * 3 + (2 - kx)) * C + c4 * 4);
                acc.x = fmaf(g, k.x, acc.x); acc.y = fmaf(g, k.y, acc.y);
                acc.z = fmaf(g, k.z, acc.z); acc.w = fmaf(g, k.w, acc.w);
            }
        }
        float4* o = reinterpret_cast<float4*>(dxp + (((size_t)b * Hp + Py) * Wp + Px) * C + c4 * 4);
        if (accumulate) {
            const float4 p = *o;
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        *o = acc;
    }
}

// partial[blk][tap][c] = sum over the block's pixels of dz * x(tap shifted); partial[blk][9*C] = sum dz
__global__ __launch_bounds__(256) void dispconv_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                             float* __restrict__ partial, int B, int H, int W, int C,
                                                             int pix_per_block) {
    __shared__ float red[37 * 256];
    const int C4 = C / 4;
    const int lanes = 256 / C4;                // pixel lanes (C4 <= 64)
    const int cq = threadIdx.x % C4, pl = threadIdx.x / C4;
    const int total = B * H * W;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(total, p0 + pix_per_block);
    float acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    if (pl < lanes) {
        for (int p = p0 + pl; p < p1; p += lanes) {
            const int xx = p % W, yy = (p / W) % H, b = p / (W * H);
            const float g = dz[p];
            if (cq == 0) bsum += g;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = reflect_idx(yy + ky - 1, H);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = reflect_idx(xx + kx - 1, W);
                    const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * H + iy) * W + ix) * C + cq * 4);
                    const int t = (ky * 3 + kx) * 4;
                    acc[t + 0] = fmaf(g, v.x, acc[t + 0]); acc[t + 1] = fmaf(g, v.y, acc[t + 1]);
                    acc[t + 2] = fmaf(g, v.z, acc[t + 2]); acc[t + 3] = fmaf(g, v.w, acc[t + 3]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) red[k * 256 + threadIdx.x] = acc[k];
    red[36 * 256 + threadIdx.x] = bsum;
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * (9 * C + 1);
    for (int e = threadIdx.x; e < 9 * C; e += 256) {
        const int tap = e / C, c = e - tap * C;
        const int q = c >> 2, comp = c & 3;
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[(tap * 4 + comp) * 256 + l * C4 + q];
        out[e] = s;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[36 * 256 + l * C4];
        out[9 * C] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// One block per pose pass n: mean over pixels of x[n,:, :256], then the 12x256 matvec, x0.01.
__global__ __launch_bounds__(256) void pose_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float* __restrict__ mean,
                                                            float* __restrict__ pose, int HW) {
    // thread = (channel quad c4, pixel lane pl): 16-byte loads, four independent pixel lanes per channel quad,
    // combined in lane order; then 16 lanes per output row for the 12x256 matvec.  (One thread per channel
    // walking all pixels was a ~30 us serial chain at the very end of the pose branch.)
    __shared__ float4 part[4][64];
    __shared__ float ms[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int c4 = tid & 63, pl = tid >> 6;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = pl; p < HW; p += 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)n * HW + p) * 256 + c4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    part[pl][c4] = s;
    __syncthreads();
    if (tid < 64) {
        float4 t = part[0][tid];
        for (int k = 1; k < 4; ++k) { const float4 v = part[k][tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        t.x /= (float)HW; t.y /= (float)HW; t.z /= (float)HW; t.w /= (float)HW;
        *reinterpret_cast<float4*>(&ms[tid * 4]) = t;
        *reinterpret_cast<float4*>(mean + (size_t)n * 256 + tid * 4) = t;
    }
    __syncthreads();
    const int o = tid >> 4, kk = tid & 15;     // 16 output groups (12 used) x 16 lanes
    float a = 0.f;
    if (o < 12)
        for (int k = kk; k < 256; k += 16) a = fmaf(w2[o * 256 + k], ms[k], a);
    a += wave_shfl_xor(a, 1); a += wave_shfl_xor(a, 2); a += wave_shfl_xor(a, 4); a += wave_shfl_xor(a, 8);
    if (o < 12 && kk == 0) pose[n * 12 + o] = 0.01f * (a + b2[o]);
}

// blocks [0,N): dz1[n,p,c] = relu'(x) * (sum_o w2[o][c] * 0.01*dpose[n][o]) / HW
// block N     : dw2[o][c] = sum_n 0.01*dpose[n][o]*mean[n][c], db2[o] = sum_n 0.01*dpose[n][o]
__global__ __launch_bounds__(256) void pose_head_bwd_kernel(const float* __restrict__ dpose, const float* __restrict__ x,
                                                            const float* __restrict__ w2, const float* __restrict__ mean,
                                                            float* __restrict__ dz1, float* __restrict__ dw2,
                                                            float* __restrict__ db2, int N, int HW, float gscale) {
    const int c = threadIdx.x;
    if ((int)blockIdx.x == N) {
        for (int o = 0; o < 12; ++o) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s = fmaf(0.01f * dpose[n * 12 + o], mean[(size_t)n * 256 + c], s);
            dw2[o * 256 + c] = s * gscale;
        }
        if (c < 12) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += 0.01f * dpose[n * 12 + c];
            db2[c] = s * gscale;
        }
        return;
    }
    const int n = blockIdx.x;
    const int c4 = c & 63, pl = c >> 6;      // channel quad, pixel lane (16-byte accesses, 4 pixels in flight)
    float dm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = 0.f;
        for (int o = 0; o < 12; ++o) v = fmaf(w2[o * 256 + c4 * 4 + k], 0.01f * dpose[n * 12 + o], v);
        dm[k] = v / (float)HW;
    }
    for (int p = pl; p < HW; p += 4) {
        const size_t i = ((size_t)n * HW + p) * 256 + c4 * 4;
        const float4 xv = *reinterpret_cast<const float4*>(x + i);
        float4 g;
        g.x = xv.x > 0.f ? dm[0] : 0.f; g.y = xv.y > 0.f ? dm[1] : 0.f;
        g.z = xv.z > 0.f ? dm[2] : 0.f; g.w = xv.w > 0.f ? dm[3] : 0.f;
        *reinterpret_cast<float4*>(dz1 + i) = g;
    }
}

}  // namespace clslam

using namespace clslam;

static int grid_for(size_t total) { return (int)std::min<size_t>(4096, (total + 255) / 256); }

extern "C" int clslam_dispconv_fwd(const float* x, const float* w, const float* bias, float* disp, int batch, int h,
                                   int wd, int ch, void* stream) {
    if (batch == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(x && w && bias && disp && (ch == 16 || ch == 32 || ch == 64 || ch == 128), "dispconv_fwd: ch must be 16/32/64/128");
    CLSLAM_REQUIRE(h >= 2 && wd >= 2, "dispconv_fwd: reflection padding needs at least 2x2 pixels");
    hipStream_t st = (hipStream_t)stream;
    // tile shapes: 22-31 KiB of LDS each, a whole number of 256-lane passes, halo overhead 1.3-1.9x
    switch (ch) {
        case 16: return launch_dispconv_fwd<16, 32, 8>(x, w, bias, disp, batch, h, wd, st);
        case 32: return launch_dispconv_fwd<32, 32, 4>(x, w, bias, disp, batch, h, wd, st);
        case 64: return launch_dispconv_fwd<64, 16, 4>(x, w, bias, disp, batch, h, wd, st);
        default: return launch_dispconv_fwd<128, 8, 4>(x, w, bias, disp, batch, h, wd, st);
    }
}

extern "C" int clslam_dispconv_bwd_data(const float* dz, const float* w, float* dxp, int batch, int h, int wd, int ch,
                                        int accumulate, void* stream) {
    CLSLAM_REQUIRE(dz && w && dxp && ch % 4 == 0 && ch <= 256, "dispconv_bwd_data: bad args");
    const size_t total = (size_t)batch * (h + 2) * (wd + 2) * (ch / 4);
    if (!total) return CLSLAM_OK;
    hipLaunchKernelGGL(dispconv_bwd_data_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dz, w, dxp,
                       batch, h, wd, ch, accumulate);
    return check_launch("dispconv_bwd_data");
}

extern "C" int clslam_dispconv_wgrad_blocks(int pixels) { return std::max(1, std::min(1024, cdiv(pixels, 128))); }

// partial: clslam_dispconv_wgrad_blocks(B*h*w) * (9*ch+1) floats; reduce with clslam_reduce_partials
// (n = 9*ch+1: the 9*ch weight gradients [tap][c] followed by the bias gradient).
extern "C" int clslam_dispconv_wgrad(const float* dz, const float* x, float* partial, int batch, int h, int wd, int ch,
                                     void* stream) {
    CLSLAM_REQUIRE(dz && x && partial && ch % 4 == 0 && ch <= 256, "dispconv_wgrad: bad args");
    const int pixels = batch * h * wd;
    const int blocks = clslam_dispconv_wgrad_blocks(pixels);
    const int ppb = cdiv(std::max(pixels, 1), blocks);
    hipLaunchKernelGGL(dispconv_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dz, x,
                       partial, batch, h, wd, ch, ppb);
    return check_launch("dispconv_wgrad");
}

extern "C" int clslam_pose_head_fwd(const float* x, const float* w2, const float* b2, float* mean, float* pose, int n,
                                    int hw, void* stream) {
    CLSLAM_REQUIRE(x && w2 && b2 && mean && pose, "pose_head_fwd: null");
    if (!n) return CLSLAM_OK;
    hipLaunchKernelGGL(pose_head_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, x, w2, b2, mean, pose, hw);
    return check_launch("pose_head_fwd");
}

extern "C" int clslam_pose_head_bwd(const float* dpose, const float* x, const float* w2, const float* mean, float* dz1,
                                    float* dw2, float* db2, int n, int hw, float grad_scale, void* stream) {
    CLSLAM_REQUIRE(dpose && x && w2 && mean && dz1 && dw2 && db2, "pose_head_bwd: null");
    hipLaunchKernelGGL(pose_head_bwd_kernel, dim3(n + 1), dim3(256), 0, (hipStream_t)stream, dpose, x, w2, mean, dz1, dw2,
                       db2, n, hw, grad_scale);
    return check_launch("pose_head_bwd");
}
