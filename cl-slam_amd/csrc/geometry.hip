// K8/K9/K13(geometry half)/K14: pose -> matrices, view synthesis and their backward.
//
// Reference: depth_pose_prediction/utils.py:34-117 (axis-angle/translation -> 4x4, `invert` for
// frame -1), networks/layers.py:51-104 (BackprojectDepth, Project3D), utils.py:120-142
// (disp_to_depth), dpp.py:986-1017 (bilinear upsample of the disparity to full resolution,
// F.grid_sample(bilinear, border, align_corners=True) of the UN-augmented scale-0 source frame with
// scale-0 intrinsics).  All of it is HBM-bound per-pixel work: one thread per pixel, planar NCHW
// images exactly as the reference's sample/output dicts hold them, coalesced along x.
#include "common.h"
#include "geometry_dev.h"

namespace clslam {

// ------------------------------------------------------------------------------------------------
// One thread per (frame fi, sample b).  pose rows n = fi*B + b hold [axis_angle(3), translation(3), ...].
__global__ void pose_to_proj_kernel(const float* __restrict__ pose, const float* __restrict__ Kmat, float* __restrict__ T,
                                    float* __restrict__ P, int B) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= 2 * B) return;
    const int fi = n / B, b = n - fi * B;
    const bool invert = (fi == 0);  // frame -1 (dpp.py:970-973)
    const float* ps = pose + (size_t)n * 12;
    const float vx = ps[0], vy = ps[1], vz = ps[2];
    float t0 = ps[3], t1 = ps[4], t2 = ps[5];
    const float angle = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = angle + 1e-7f;
    const float x = vx / inv, y = vy / inv, z = vz / inv;
    const float ca = cosf(angle), sa = sinf(angle), C = 1.f - ca;
    const float xs = x * sa, ys = y * sa, zs = z * sa;
    const float xC = x * C, yC = y * C, zC = z * C;
    const float xyC = x * yC, yzC = y * zC, zxC = z * xC;
    float R[3][3] = {{x * xC + ca, xyC - zs, zxC + ys}, {xyC + zs, y * yC + ca, yzC - xs}, {zxC - ys, yzC + xs, z * zC + ca}};
    float M[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) M[i][j] = (i == j) ? 1.f : 0.f;
    if (invert) {
        t0 = -t0; t1 = -t1; t2 = -t2;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[i][j] = R[j][i];
            M[i][3] = R[0][i] * t0 + R[1][i] * t1 + R[2][i] * t2;
        }
    } else {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[i][j] = R[i][j];
        }
        M[0][3] = t0; M[1][3] = t1; M[2][3] = t2;
    }
    float* To = T + (size_t)n * 16;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) To[i * 4 + j] = M[i][j];
    const float* K = Kmat + (size_t)b * 16;
    float* Po = P + (size_t)n * 12;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s = fmaf(K[i * 4 + k], M[k][j], s);
            Po[i * 4 + j] = s;
        }
}

// ------------------------------------------------------------------------------------------------
// depth[s,b,y,x] and warped[s,fi,b,c,y,x] for the pyr.n scales of a pyramid (one launch).
// The kernel is VALU-bound, not HBM-bound (94 MB per launch at B = 5): its first version spent ~880 issue slots per pixel,
// a third of them on index arithmetic (a flat index split by runtime divisors, 64-bit address chains, quarter-rate 32-bit
// multiplies).  Now a block owns a 4 x 64 pixel tile of one (sample, scale): the tile origin is scalar, every address is a
// uniform base pointer plus one unsigned 32-bit offset (saddr + voffset form) and products are 24-bit (full rate).  The
// floating-point expressions are unchanged.
constexpr int WF_TH = 4, WF_TW = 64;
__global__ __launch_bounds__(256) void warp_fwd_kernel(Pyramid pyr, const float* __restrict__ src_m1,
                                                       const float* __restrict__ src_p1, const float* __restrict__ Kinv,
                                                       const float* __restrict__ P, float* __restrict__ depth_all,
                                                       float* __restrict__ warped_all, int B, int H, int W, float da, float db,
                                                       int dmode, int tilesX) {
    const int b = blockIdx.y, sc = blockIdx.z + pyr.base;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x = tx * WF_TW + (int)(threadIdx.x & 63), y = ty * WF_TH + (int)(threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const unsigned HW = (unsigned)(H * W);
    const unsigned pix = __umul24((unsigned)y, (unsigned)W) + (unsigned)x;
    const int h = pyr.h[sc], w = pyr.w[sc];
    const float disp = upsample_disp(pyr.disp[sc] + (size_t)b * h * w, h, w, H, W, y, x);
    const float dep = disp_to_depth_dev(disp, da, db, dmode);
    (depth_all + ((size_t)sc * B + b) * HW)[pix] = dep;
    const float* Ki = Kinv + (size_t)b * 16;
    const float fx = (float)x, fy = (float)y;
    float X[3], cam[3];
    backproject_px(Ki, fx, fy, dep, cam, X);
    // The 12 bilinear taps of a frame are loaded unconditionally from clamped addresses and out-of-image taps get weight 0
    // (adding 0 leaves the sum bit-identical): with `if (x1ok) v += pl[..]` every tap was a branch + a load + a wait.
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        const float* Pm = P + ((size_t)fi * B + b) * 12;
        float u, v, den;
        project_px(Pm, X, u, v, den);
        const Sample s = sample_coords(u, v, H, W);
        const float wx1 = s.ix - (float)s.x0, wy1 = s.iy - (float)s.y0;
        const float wx0 = (float)(s.x0 + 1) - s.ix, wy0 = (float)(s.y0 + 1) - s.iy;
        const bool x1ok = s.x0 + 1 < W, y1ok = s.y0 + 1 < H;
        const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
        const unsigned o00 = __umul24((unsigned)s.y0, (unsigned)W) + (unsigned)s.x0;
        const unsigned o10 = o00 + (x1ok ? 1u : 0u);
        const unsigned o01 = o00 + (y1ok ? (unsigned)W : 0u);
        const unsigned o11 = o01 + (x1ok ? 1u : 0u);
        const float* src = (fi == 0 ? src_m1 : src_p1) + (size_t)b * 3 * HW;
        float* wout = warped_all + (((size_t)sc * 2 + fi) * B + b) * 3 * HW;
        float nw[3], ne[3], sw[3], se[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pl = src + (size_t)c * HW;
            nw[c] = pl[o00]; ne[c] = pl[o10]; sw[c] = pl[o01]; se[c] = pl[o11];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = nw[c] * w00;
            if (x1ok) v += ne[c] * w10;
            if (y1ok) v += sw[c] * w01;
            if (x1ok && y1ok) v += se[c] * w11;
            (wout + (size_t)c * HW)[pix] = v;
        }
    }
}

// Backward of warp_fwd for one scale.  dpred[fi,b,c,y,x] = dL/d warped.  Writes
//   ddisp_up[b,y,x] = dL/d(upsampled disparity)   and   dP_partial[b][blk][fi*12+k] (block sums).
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ disp_s,
                                                       int h, int w, const float* __restrict__ src_m1,
                                                       const float* __restrict__ src_p1, const float* __restrict__ Kinv,
                                                       const float* __restrict__ P, float* __restrict__ ddisp_up,
                                                       double* __restrict__ dP_partial, int B, int H, int W, float da, float db,
                                                       int dmode, int pix_per_block) {
    __shared__ double red[4][24];
    const int b = blockIdx.y;
    const int HW = H * W;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    double dPacc[24];       // (stand-alone form for the kernel-level tests: accumulators in double from the first term)
#pragma unroll
    for (int k = 0; k < 24; ++k) dPacc[k] = 0.0;
    const float* Ki = Kinv + (size_t)b * 16;
    for (int pi = p0 + (int)threadIdx.x; pi < p1; pi += 256) {
        const int x = pi % W, y = pi / W;
        const float disp = upsample_disp(disp_s + (size_t)b * h * w, h, w, H, W, y, x);
        const float dep = disp_to_depth_dev(disp, da, db, dmode);
        const float fx = (float)x, fy = (float)y;
        float cam[3], X[3];
        backproject_px(Ki, fx, fy, dep, cam, X);
        float ddepth = 0.f;
        for (int fi = 0; fi < 2; ++fi) {
            const float* Pm = P + ((size_t)fi * B + b) * 12;
            float u, v, den;
            project_px(Pm, X, u, v, den);
            const Sample s = sample_coords(u, v, H, W);
            const float wx1 = s.ix - (float)s.x0, wy1 = s.iy - (float)s.y0;
            const float wx0 = (float)(s.x0 + 1) - s.ix, wy0 = (float)(s.y0 + 1) - s.iy;
            const bool x1ok = s.x0 + 1 < W, y1ok = s.y0 + 1 < H;
            const float* src = (fi == 0 ? src_m1 : src_p1) + (size_t)b * 3 * HW;
            float gix = 0.f, giy = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float g = dpred[(((size_t)fi * B + b) * 3 + c) * HW + pi];
                const float* pl = src + (size_t)c * HW;
                const float nw = pl[s.y0 * W + s.x0];
                const float ne = x1ok ? pl[s.y0 * W + s.x0 + 1] : 0.f;
                const float sw = y1ok ? pl[(s.y0 + 1) * W + s.x0] : 0.f;
                const float se = (x1ok && y1ok) ? pl[(s.y0 + 1) * W + s.x0 + 1] : 0.f;
                gix += g * (-nw * wy0 + ne * wy0 - sw * wy1 + se * wy1);
                giy += g * (-nw * wx0 - ne * wx1 + sw * wx0 + se * wx1);
            }
            const float du = gix * s.mx, dv = giy * s.my;
            float dp[3];
            dp[0] = du / den;
            dp[1] = dv / den;
            dp[2] = -(du * u + dv * v) / den;
            for (int i = 0; i < 3; ++i) {
                dPacc[fi * 12 + i * 4 + 0] += dp[i] * X[0];
                dPacc[fi * 12 + i * 4 + 1] += dp[i] * X[1];
                dPacc[fi * 12 + i * 4 + 2] += dp[i] * X[2];
                dPacc[fi * 12 + i * 4 + 3] += dp[i];
            }
            ddepth += ddepth_from_duv(Pm, cam, du, dv, 1.f / den);
        }
        float dd;
        if (dmode == 2) dd = -db * dep * dep * ddepth;
        else dd = -dep / disp * ddepth;
        ddisp_up[(size_t)b * HW + pi] = dd;
    }
    // block reduction of the 24 dP entries (fixed order -> deterministic)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const double s = wave_sum_f64(dPacc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 24)
        dP_partial[((size_t)b * gridDim.x + blockIdx.x) * 24 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// One block per sample b.  Phase 1: dP[fi][12] = sum over (scale, block) partials (40 row-lanes x 6 column
// quads, fixed order).  Phase 2 (threads 0,1 = frame idx): the pose chain backward (autograd of
// utils.py:34-117 and layers.py:94) plus the velocity-loss gradient (dpp.py:1125-1146).
// EVERYTHING HERE IS DOUBLE: the partials arrive as doubles (wave_sum_f64 in the loss backward), and dM = K^T dP below
// subtracts products of |K| ~ 300-600 whose sum is orders of magnitude smaller (dP[2] is -(u dP[0] + v dP[1]) pixel by
// pixel, u ~ cx) -- in fp32 this one spot put the pose-decoder gradients of the B = 5 benchmark step 4.5x further from
// the float64 gradient than torch's own fp32 autograd (7.4e-4 vs 1.6e-4, profiles/r03_backward_parity.txt).  24 numbers
// per sample: the precision is free.
__global__ __launch_bounds__(256) void pose_bwd_kernel(const double* __restrict__ dP_partial, int nscale, int nblk,
                                                       const float* __restrict__ pose, const float* __restrict__ Kmat,
                                                       const double* __restrict__ dist0, const double* __restrict__ dist1,
                                                       const float* __restrict__ sample_w, float vel_scale,
                                                       float* __restrict__ dpose, int B) {
    constexpr int RL = 40;                 // row lanes: 40 x 6 quads of doubles = 240 threads
    __shared__ double red[RL][24];
    __shared__ double dPs[24];
    const int b = blockIdx.x;
    {
        // 32-byte rows of four doubles, 40 independent row lanes (a 10-lane scalar version was a ~25 us chain of dependent-
        // latency loads: 960 partial rows per sample at 192x640)
        const int q = threadIdx.x % 6, rl = threadIdx.x / 6;
        if (rl < RL) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int sc = 0; sc < nscale; ++sc) {
                const double* base = dP_partial + ((size_t)sc * B + b) * nblk * 24 + 4 * q;
                for (int blk = rl; blk < nblk; blk += RL) {
                    const double* r = base + (size_t)blk * 24;      // (the compiler merges the four into two 16-byte loads)
                    s0 += r[0]; s1 += r[1]; s2 += r[2]; s3 += r[3];
                }
            }
            red[rl][q * 4 + 0] = s0; red[rl][q * 4 + 1] = s1; red[rl][q * 4 + 2] = s2; red[rl][q * 4 + 3] = s3;
        }
    }
    __syncthreads();
    if (threadIdx.x < 24) {
        double s = 0.0;
        for (int r = 0; r < RL; ++r) s += red[r][threadIdx.x];
        dPs[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x >= 2) return;
    const int fi = threadIdx.x;
    const int n = fi * B + b;
    double dP[12];
    for (int k = 0; k < 12; ++k) dP[k] = dPs[fi * 12 + k];
    const float* K = Kmat + (size_t)b * 16;
    // dM[k][j] = sum_{i<3} K[i][k] * dP[i][j]
    double dM[4][4];
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j)
            dM[k][j] = (double)K[0 * 4 + k] * dP[0 * 4 + j] + (double)K[1 * 4 + k] * dP[1 * 4 + j] + (double)K[2 * 4 + k] * dP[2 * 4 + j];
    const float* ps = pose + (size_t)n * 12;
    const double vx = ps[0], vy = ps[1], vz = ps[2];
    const double t[3] = {(double)ps[3], (double)ps[4], (double)ps[5]};
    const double angle = sqrt(vx * vx + vy * vy + vz * vz);
    const double inv = angle + 1e-7;
    const double x = vx / inv, y = vy / inv, z = vz / inv;
    const double ca = cos(angle), sa = sin(angle), C = 1.0 - ca;
    const double R[3][3] = {{x * x * C + ca, x * y * C - z * sa, z * x * C + y * sa},
                            {x * y * C + z * sa, y * y * C + ca, y * z * C - x * sa},
                            {z * x * C - y * sa, y * z * C + x * sa, z * z * C + ca}};
    double G[3][3], dt[3];
    if (fi == 0) {  // inverted: M3 = R^T, Mt[i] = -sum_k R[k][i] t[k]
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) G[i][j] = dM[j][i];
        for (int k = 0; k < 3; ++k) {
            double s = 0.0;
            for (int i = 0; i < 3; ++i) {
                G[k][i] += -t[k] * dM[i][3];
                s += R[k][i] * dM[i][3];
            }
            dt[k] = -s;
        }
    } else {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) G[i][j] = dM[i][j];
            dt[i] = dM[i][3];
        }
    }
    const double xC = x * C, yC = y * C, zC = z * C;
    const double s01 = G[0][1] + G[1][0], s02 = G[0][2] + G[2][0], s12 = G[1][2] + G[2][1];
    const double dx = G[0][0] * 2.0 * xC + s01 * yC + s02 * zC + (G[2][1] - G[1][2]) * sa;
    const double dy = G[1][1] * 2.0 * yC + s01 * xC + s12 * zC + (G[0][2] - G[2][0]) * sa;
    const double dz = G[2][2] * 2.0 * zC + s02 * xC + s12 * yC + (G[1][0] - G[0][1]) * sa;
    const double dC = G[0][0] * x * x + G[1][1] * y * y + G[2][2] * z * z + s01 * x * y + s02 * z * x + s12 * y * z;
    const double dca = G[0][0] + G[1][1] + G[2][2] - dC;
    const double dsa = (G[1][0] - G[0][1]) * z + (G[0][2] - G[2][0]) * y + (G[2][1] - G[1][2]) * x;
    double dtheta = -sa * dca + ca * dsa;
    dtheta += -(dx * vx + dy * vy + dz * vz) / (inv * inv);
    double dv[3] = {dx / inv, dy / inv, dz / inv};
    if (angle > 0.0) { dv[0] += dtheta * vx / angle; dv[1] += dtheta * vy / angle; dv[2] += dtheta * vz / angle; }
    // velocity loss: frame idx 0 (translation 0->-1) pairs with relative_distance(0), idx 1 with (1)
    if (vel_scale > 0.f) {
        const double gt = fabs(fi == 0 ? dist0[b] : dist1[b]);
        const float nrm = sqrtf(ps[3] * ps[3] + ps[4] * ps[4] + ps[5] * ps[5]);
        const double diff = (double)nrm - gt;
        const double sg = diff > 0.0 ? 1.0 : (diff < 0.0 ? -1.0 : 0.0);
        if (nrm > 0.f) {
            const double coef = (double)sample_w[b] * (double)vel_scale * 0.5 * sg / (double)nrm;
            dt[0] += coef * t[0]; dt[1] += coef * t[1]; dt[2] += coef * t[2];
        }
    }
    float* o = dpose + (size_t)n * 12;
    o[0] = (float)dv[0]; o[1] = (float)dv[1]; o[2] = (float)dv[2];
    o[3] = (float)dt[0]; o[4] = (float)dt[1]; o[5] = (float)dt[2];
    for (int k = 6; k < 12; ++k) o[k] = 0.f;
}

}  // namespace clslam

using namespace clslam;

extern "C" int clslam_pose_to_proj(const float* pose, const float* kmat, float* cam_t_cam, float* proj, int batch,
                                   void* stream) {
    CLSLAM_REQUIRE(pose && kmat && cam_t_cam && proj, "pose_to_proj: null");
    if (!batch) return CLSLAM_OK;
    hipLaunchKernelGGL(pose_to_proj_kernel, dim3(cdiv(2 * batch, 64)), dim3(64), 0, (hipStream_t)stream, pose, kmat,
                       cam_t_cam, proj, batch);
    return check_launch("pose_to_proj");
}

extern "C" int clslam_warp_fwd(const float* disp_s, int h, int w, const float* src_m1, const float* src_p1,
                               const float* inv_k, const float* proj, float* depth, float* warped, int batch, int H, int W,
                               float min_depth, float max_depth, void* stream) {
    CLSLAM_REQUIRE(disp_s && src_m1 && src_p1 && inv_k && proj && depth && warped, "warp_fwd: null");
    CLSLAM_REQUIRE(!(min_depth <= 0.f && max_depth > 0.f), "warp_fwd: min_depth is None");
    float a, b; int mode;
    depth_mode(min_depth, max_depth, &a, &b, &mode);
    const size_t total = (size_t)batch * H * W;
    if (!total) return CLSLAM_OK;
    Pyramid pyr;
    pyr.n = 1; pyr.base = 0; pyr.disp[0] = disp_s; pyr.h[0] = h; pyr.w[0] = w;
    for (int k = 1; k < 4; ++k) { pyr.disp[k] = nullptr; pyr.h[k] = pyr.w[k] = 0; }
    const int tilesX = cdiv(W, WF_TW);
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(tilesX * cdiv(H, WF_TH), batch, 1), dim3(256), 0,
                       (hipStream_t)stream, pyr, src_m1, src_p1, inv_k, proj, depth, warped, batch, H, W, a, b, mode, tilesX);
    return check_launch("warp_fwd");
}

// All four scales in one launch: disp[s] (B,H>>s,W>>s); depth (4,B,H,W); warped (4,2,B,3,H,W).
extern "C" int clslam_warp_fwd_pyramid_range(const float* const* disp, const float* src_m1, const float* src_p1, const float* inv_k,
                                             const float* proj, float* depth, float* warped, int batch, int H, int W,
                                             float min_depth, float max_depth, int scale_lo, int scale_count, void* stream) {
    CLSLAM_REQUIRE(disp && src_m1 && src_p1 && inv_k && proj && depth && warped, "warp_fwd_pyramid: null");
    CLSLAM_REQUIRE(!(min_depth <= 0.f && max_depth > 0.f), "warp_fwd_pyramid: min_depth is None");
    CLSLAM_REQUIRE(scale_lo >= 0 && scale_count >= 0 && scale_lo + scale_count <= 4, "warp_fwd_pyramid: scales [%d, %d) outside the pyramid",
                   scale_lo, scale_lo + scale_count);
    float a, b; int mode;
    depth_mode(min_depth, max_depth, &a, &b, &mode);
    Pyramid pyr;
    pyr.n = 4; pyr.base = scale_lo;
    for (int k = 0; k < 4; ++k) { pyr.disp[k] = disp[k]; pyr.h[k] = H >> k; pyr.w[k] = W >> k; }
    const size_t total = (size_t)4 * batch * H * W;
    if (!total || !scale_count) return CLSLAM_OK;
    CLSLAM_REQUIRE(total < ((size_t)1 << 31), "warp_fwd_pyramid: batch too large for 32-bit indexing");
    const int tilesX = cdiv(W, WF_TW);
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(tilesX * cdiv(H, WF_TH), batch, scale_count), dim3(256), 0,
                       (hipStream_t)stream, pyr, src_m1, src_p1, inv_k, proj, depth, warped, batch, H, W, a, b, mode, tilesX);
    return check_launch("warp_fwd_pyramid");
}

extern "C" int clslam_warp_fwd_pyramid(const float* const* disp, const float* src_m1, const float* src_p1, const float* inv_k,
                                       const float* proj, float* depth, float* warped, int batch, int H, int W,
                                       float min_depth, float max_depth, void* stream) {
    return clslam_warp_fwd_pyramid_range(disp, src_m1, src_p1, inv_k, proj, depth, warped, batch, H, W, min_depth, max_depth, 0, 4, stream);
}

// Diagnostic twin of warp_fwd_kernel (tests/test_backward_parity.py): the bilinear CELL and the border-clip flags the
// path uses for every (scale, source frame, sample, pixel), packed as x0 | y0 << 12 | (mx != 0) << 24 | (my != 0) << 25.
// The same expressions, in the same order, as warp_fwd_kernel / the loss backward: the oracle is re-run on exactly these
// cells to attribute the part of the gradient residual that comes from samples landing on the other side of a kink.
__global__ __launch_bounds__(256) void warp_cells_kernel(Pyramid pyr, const float* __restrict__ Kinv,
                                                         const float* __restrict__ P, int* __restrict__ cells, int B, int H,
                                                         int W, float da, float db, int dmode, int tilesX) {
    const int b = blockIdx.y, sc = blockIdx.z;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x = tx * WF_TW + (int)(threadIdx.x & 63), y = ty * WF_TH + (int)(threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const unsigned HW = (unsigned)(H * W);
    const unsigned pix = __umul24((unsigned)y, (unsigned)W) + (unsigned)x;
    const int h = pyr.h[sc], w = pyr.w[sc];
    const float disp = upsample_disp(pyr.disp[sc] + (size_t)b * h * w, h, w, H, W, y, x);
    const float dep = disp_to_depth_dev(disp, da, db, dmode);
    const float* Ki = Kinv + (size_t)b * 16;
    const float fx = (float)x, fy = (float)y;
    float X[3], cam[3];
    backproject_px(Ki, fx, fy, dep, cam, X);
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        const float* Pm = P + ((size_t)fi * B + b) * 12;
        float u, v, den;
        project_px(Pm, X, u, v, den);
        const Sample s = sample_coords(u, v, H, W);
        (cells + (((size_t)sc * 2 + fi) * B + b) * HW)[pix] =
            s.x0 | (s.y0 << 12) | ((s.mx != 0.f ? 1 : 0) << 24) | ((s.my != 0.f ? 1 : 0) << 25);
    }
}

extern "C" int clslam_warp_cells_pyramid(const float* const* disp, const float* inv_k, const float* proj, int* cells, int batch,
                                         int H, int W, float min_depth, float max_depth, void* stream) {
    CLSLAM_REQUIRE(disp && inv_k && proj && cells, "warp_cells_pyramid: null");
    CLSLAM_REQUIRE(!(min_depth <= 0.f && max_depth > 0.f), "warp_cells_pyramid: min_depth is None");
    CLSLAM_REQUIRE(H < 4096 && W < 4096, "warp_cells_pyramid: image too large for the packed cell format");
    float a, b; int mode;
    depth_mode(min_depth, max_depth, &a, &b, &mode);
    Pyramid pyr;
    pyr.n = 4; pyr.base = 0;
    for (int k = 0; k < 4; ++k) { pyr.disp[k] = disp[k]; pyr.h[k] = H >> k; pyr.w[k] = W >> k; }
    if (!batch) return CLSLAM_OK;
    const int tilesX = cdiv(W, WF_TW);
    hipLaunchKernelGGL(warp_cells_kernel, dim3(tilesX * cdiv(H, WF_TH), batch, 4), dim3(256), 0, (hipStream_t)stream, pyr, inv_k,
                       proj, cells, batch, H, W, a, b, mode, tilesX);
    return check_launch("warp_cells_pyramid");
}

// Second diagnostic twin (tests/test_warp_positions.py): the sampling POSITION itself, (ix, iy) in pixels after the border
// clip, for every (scale, source frame, sample, pixel) -- the quantity the warped-image tolerance is derived from.
__global__ __launch_bounds__(256) void warp_coords_kernel(Pyramid pyr, const float* __restrict__ Kinv,
                                                          const float* __restrict__ P, float2* __restrict__ coords, int B, int H,
                                                          int W, float da, float db, int dmode, int tilesX) {
    const int b = blockIdx.y, sc = blockIdx.z;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int x = tx * WF_TW + (int)(threadIdx.x & 63), y = ty * WF_TH + (int)(threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const unsigned HW = (unsigned)(H * W);
    const unsigned pix = __umul24((unsigned)y, (unsigned)W) + (unsigned)x;
    const int h = pyr.h[sc], w = pyr.w[sc];
    const float disp = upsample_disp(pyr.disp[sc] + (size_t)b * h * w, h, w, H, W, y, x);
    const float dep = disp_to_depth_dev(disp, da, db, dmode);
    const float* Ki = Kinv + (size_t)b * 16;
    float X[3], cam[3];
    backproject_px(Ki, (float)x, (float)y, dep, cam, X);
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        float u, v, den;
        project_px(P + ((size_t)fi * B + b) * 12, X, u, v, den);
        const Sample s = sample_coords(u, v, H, W);
        (coords + (((size_t)sc * 2 + fi) * B + b) * HW)[pix] = make_float2(s.ix, s.iy);
    }
}

extern "C" int clslam_warp_coords_pyramid(const float* const* disp, const float* inv_k, const float* proj, float* coords,
                                          int batch, int H, int W, float min_depth, float max_depth, void* stream) {
    CLSLAM_REQUIRE(disp && inv_k && proj && coords, "warp_coords_pyramid: null");
    CLSLAM_REQUIRE(!(min_depth <= 0.f && max_depth > 0.f), "warp_coords_pyramid: min_depth is None");
    float a, b; int mode;
    depth_mode(min_depth, max_depth, &a, &b, &mode);
    Pyramid pyr;
    pyr.n = 4; pyr.base = 0;
    for (int k = 0; k < 4; ++k) { pyr.disp[k] = disp[k]; pyr.h[k] = H >> k; pyr.w[k] = W >> k; }
    if (!batch) return CLSLAM_OK;
    const int tilesX = cdiv(W, WF_TW);
    hipLaunchKernelGGL(warp_coords_kernel, dim3(tilesX * cdiv(H, WF_TH), batch, 4), dim3(256), 0, (hipStream_t)stream, pyr, inv_k,
                       proj, reinterpret_cast<float2*>(coords), batch, H, W, a, b, mode, tilesX);
    return check_launch("warp_coords_pyramid");
}

extern "C" int clslam_warp_bwd_blocks(int H, int W) { return std::max(1, std::min(256, cdiv(H * W, 1024))); }

extern "C" int clslam_warp_bwd(const float* dpred, const float* disp_s, int h, int w, const float* src_m1,
                               const float* src_p1, const float* inv_k, const float* proj, float* ddisp_up, double* dp_partial,
                               int batch, int H, int W, float min_depth, float max_depth, void* stream) {
    CLSLAM_REQUIRE(dpred && disp_s && src_m1 && src_p1 && inv_k && proj && ddisp_up && dp_partial, "warp_bwd: null");
    float a, b; int mode;
    depth_mode(min_depth, max_depth, &a, &b, &mode);
    if (!batch) return CLSLAM_OK;
    const int nblk = clslam_warp_bwd_blocks(H, W);
    const int ppb = cdiv(H * W, nblk);
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(nblk, batch), dim3(256), 0, (hipStream_t)stream, dpred, disp_s, h, w, src_m1,
                       src_p1, inv_k, proj, ddisp_up, dp_partial, batch, H, W, a, b, mode, ppb);
    return check_launch("warp_bwd");
}

extern "C" int clslam_pose_bwd(const double* dp_partial, int nscale, int nblk, const float* pose, const float* kmat,
                               const double* dist0, const double* dist1, const float* sample_w, float vel_scale,
                               float* dpose, int batch, void* stream) {
    CLSLAM_REQUIRE(dp_partial && pose && kmat && dpose && sample_w, "pose_bwd: null");
    CLSLAM_REQUIRE(vel_scale <= 0.f || (dist0 && dist1), "pose_bwd: distances missing");
    if (!batch) return CLSLAM_OK;
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, dp_partial, nscale, nblk,
                       pose, kmat, dist0, dist1, sample_w, vel_scale, dpose, batch);
    return check_launch("pose_bwd");
}
