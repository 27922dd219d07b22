// Device helpers shared by geometry.hip and loss.hip: bilinear disparity upsampling, disp->depth,
// projected-pixel -> sampling coordinates (reference dpp.py:988-995, utils.py:120-142,
// networks/layers.py:93-104 + F.grid_sample(border, align_corners=True)).
#pragma once
#include "common.h"

namespace clslam {

// up to four disparity maps of a pyramid (scale s has h[s] x w[s] pixels, batch-major)
struct Pyramid {
    const float* disp[4];
    int h[4], w[4];
    int n;
    int base;      // first scale of THIS launch (warp_fwd / loss_bwd2 / disp_grad take a sub-range of the pyramid: grid index + base)
};

// EVERY function between a disparity and a bilinear cell below has floating-point contraction switched off.  The forward
// (warp_fwd), the loss backward and the diagnostic read-out each re-derive the sampling position of a pixel, and the
// backward differentiates the cell the forward sampled: the three must agree to the BIT.  With hipcc free to fuse a
// multiply-add in one kernel and not in another they did not -- for a sample within one ulp of a cell boundary
// (u = 447.99998) the backward floored to the next cell and differentiated the wrong pair of pixels: two isolated pixels per
// scale at 192x640, B = 5, each off by about its own magnitude (tools/diag_bwd.py).
__device__ __forceinline__ float upsample_disp(const float* __restrict__ d, int h, int w, int H, int W, int y, int x) {
#pragma clang fp contract(off)
    // F.interpolate(..., mode='bilinear', align_corners=False): src = (dst+0.5)*in/out - 0.5, clamped at 0
    const float ry = (float)h / (float)H, rx = (float)w / (float)W;
    float sy = ry * ((float)y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
    float sx = rx * ((float)x + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    return hy * (hx * d[y0 * w + x0] + lx * d[y0 * w + x1]) + ly * (hx * d[y1 * w + x0] + lx * d[y1 * w + x1]);
}

__device__ __forceinline__ float disp_to_depth_dev(float disp, float dmin_a, float dmin_b, int mode) {
#pragma clang fp contract(off)
    // mode 0: 1/disp; 1: min_depth/disp (a = min_depth); 2: 1/(a + b*disp) (a = 1/max, b = 1/min - 1/max)
    if (mode == 0) return 1.f / disp;
    if (mode == 1) return dmin_a / disp;
    return 1.f / (dmin_a + dmin_b * disp);
}

struct Sample {
    float ix, iy;       // clipped pixel coordinates
    float mx, my;       // gradient multipliers of the clip (0 at / outside the border)
    int x0, y0;         // floor
};

__device__ __forceinline__ Sample sample_coords(float u, float v, int H, int W) {
#pragma clang fp contract(off)
    // Project3D normalisation (layers.py:101-103) followed by grid_sample's un-normalisation
    // (align_corners=True) and border clipping.
    Sample s;
    const float gx = (u / (float)(W - 1) - 0.5f) * 2.f;
    const float gy = (v / (float)(H - 1) - 0.5f) * 2.f;
    float ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
    float iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    s.mx = 1.f; s.my = 1.f;
    if (!(ix > 0.f)) { ix = 0.f; s.mx = 0.f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); s.mx = 0.f; }
    if (!(iy > 0.f)) { iy = 0.f; s.my = 0.f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); s.my = 0.f; }
    s.ix = ix; s.iy = iy;
    s.x0 = (int)floorf(ix); s.y0 = (int)floorf(iy);
    return s;
}


// cam = K^-1 [x, y, 1]^T (row-major 4x4 Ki), X = depth * cam
__device__ __forceinline__ void backproject_px(const float* __restrict__ Ki, float fx, float fy, float dep, float* cam, float* X) {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < 3; ++i) { cam[i] = Ki[i * 4 + 0] * fx + Ki[i * 4 + 1] * fy + Ki[i * 4 + 2]; X[i] = dep * cam[i]; }
}

// p = Pm[:, :3] X + Pm[:, 3] (row-major 3x4), den = p2 + 1e-7, (u, v) = (p0, p1) / den  (layers.py:93-104)
__device__ __forceinline__ void project_px(const float* __restrict__ Pm, const float* X, float& u, float& v, float& den) {
#pragma clang fp contract(off)
    float p[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = Pm[i * 4 + 0] * X[0] + Pm[i * 4 + 1] * X[1] + Pm[i * 4 + 2] * X[2] + Pm[i * 4 + 3];
    den = p[2] + 1e-7f;
    u = p[0] / den;
    v = p[1] / den;
}

// dL/d depth of one source frame from (du, dv) = dL/d(u, v), the projection row-major Pm (3x4), cam = K^-1 [x, y, 1] and
// 1/den.  With p = depth * (Pm[:, :3] cam) + Pm[:, 3] and u = p0 / den, den = p2 + 1e-7:
//     d u / d depth = (a0 * (P23 + 1e-7) - a2 * P03) / den^2,   a = Pm[:, :3] cam      (v alike),
// i.e. the form in which the `u * du` parts of the chain rule's three terms have cancelled ANALYTICALLY.  Summing the three
// terms in fp32 (what autograd does) cancels them numerically to ~1e-7 * |u| * |du| -- with this network's small
// translations that is as large as the result itself at |u| of a few hundred pixels (tests/test_backward_parity.py:
// isolated pixels off by a factor of two against the float64 oracle).
__device__ __forceinline__ float ddepth_from_duv(const float* __restrict__ Pm, const float* cam, float du, float dv,
                                                 float inv_den) {
    const float a0 = Pm[0] * cam[0] + Pm[1] * cam[1] + Pm[2] * cam[2];
    const float a1 = Pm[4] * cam[0] + Pm[5] * cam[1] + Pm[6] * cam[2];
    const float a2 = Pm[8] * cam[0] + Pm[9] * cam[1] + Pm[10] * cam[2];
    const float p23 = Pm[11] + 1e-7f;
    return (du * (a0 * p23 - a2 * Pm[3]) + dv * (a1 * p23 - a2 * Pm[7])) * inv_den * inv_den;
}

// mode/a/b of disp_to_depth_dev from (min_depth, max_depth); values <= 0 stand for None (utils.py:120-142)
inline void depth_mode(float min_depth, float max_depth, float* a, float* b, int* mode) {
    if (min_depth <= 0.f && max_depth <= 0.f) { *mode = 0; *a = 0.f; *b = 0.f; }
    else if (max_depth <= 0.f) { *mode = 1; *a = min_depth; *b = 0.f; }
    else { *mode = 2; *a = 1.f / max_depth; *b = 1.f / min_depth - 1.f / max_depth; }
}

}  // namespace clslam
