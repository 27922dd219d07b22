// K10/K11/K12/K13(photometric half): SSIM + L1 photometric loss with auto-masking /
// min-reprojection, the smoothness term (reference behaviour incl. its flattening quirk), the
// velocity term, and the backward of all of it down to dL/d(warped image) and dL/d(sigmoid input).
//
// Reference: networks/layers.py:107-137 (SSIM: 3x3 box means on a ReflectionPad2d(1) image),
// dpp.py:1178-1192 (0.85*mean_c SSIM + 0.15*mean_c L1), dpp.py:1038-1076 (identity losses + 1e-5
// noise, min over [id(-1), id(+1), reproj(-1), reproj(+1)], mean over W then H, sample weights),
// dpp.py:1084-1101 (mean-normalised disparity, smoothness x 1e-3/2^s, sum/4),
// dpp.py:1148-1176 (smoothness: masked_select flattens the batch, so term i is the single flat
// element i of sample 0's gradient maps -- SURVEY.md 0.3), dpp.py:1105-1112 and 1125-1146
// (velocity), autograd for the backward.  HBM-bound stencil / pointwise kernels on planar images.
#include "common.h"
#include "geometry_dev.h"

namespace clslam {

__device__ __forceinline__ int refl(int i, int n) { return reflect_idx(i, n); }

// x / 9 correctly rounded in 3 instructions (Markstein: q = RN(x * r), q' = RN(q + RN(x - 9q) * r) with
// r = RN(1/9)) instead of the ~10 of an IEEE division.  The 3x3 window means feed sigma = E[x^2] - mu^2, a
// cancellation that amplifies a 1-ulp difference in the means to ~1e-5 in the SSIM value -- enough to flip
// near-ties of the 4-way min against the reference, so the quotient has to be the exact one.
__device__ __forceinline__ float div9(float x) {
    const float r = 1.f / 9.f;
    const float q = x * r;
    return fmaf(fmaf(-9.f, q, x), r, q);
}
__device__ __forceinline__ f32x2 div9(f32x2 x) {
    const f32x2 r = {1.f / 9.f, 1.f / 9.f}, m9 = {-9.f, -9.f};
    const f32x2 q = x * r;
    return pk_fma(pk_fma(m9, q, x), r, q);
}

// Fused photometric map + auto-masking for the whole pyramid (one launch, grid (nblk, B, nscale)):
// both reprojection maps of a pixel are evaluated in registers, the 4-way min / argmin is taken against
// the (scale-independent) identity maps + noise, and ONLY the selected frame's 9 SSIM coefficients are
// stored (coef_sel (S,B,9,H,W)); no reprojection maps, no per-frame coefficient planes in HBM.
constexpr int PA_TH = 8, PA_TW = 64, PA_PH = PA_TH + 2, PA_PW = PA_TW + 2;

// LDS-tiled: a block owns an 8 x 64 pixel tile of one (scale, sample).  The target tile and both warped
// tiles (3 channels each, 1-pixel halo with the reflection padding of the SSIM blocks resolved while
// staging) are read from HBM once (x1.29 halo) instead of 9x through L1 per window; the 3x3 statistics
// then come from LDS.  TRAIN: coef_sel_all != nullptr (a runtime pointer select would push the
// coefficient arrays to scratch).
// Tie-break noise of dpp.py:1055-1056 (`identity_reprojection_losses += randn(...) * 1e-5`) drawn inside the kernel:
// Philox4x32-10 keyed by the caller's seed, counter = (element index, draw offset of the step) -> two N(0,1) samples
// by Box-Muller.  The reference's generator is torch's global one on the compute device; its stream is not
// reproducible across devices either, parity runs inject captured tensors instead (set_tie_break_noise).
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// element e of the draw `offset`: noise for the two identity channels of one pixel
__device__ __forceinline__ void tie_break_pair(unsigned long long seed, unsigned long long offset, unsigned long long e,
                                               float& n0, float& n1) {
    unsigned r[4];
    philox4x32_10((unsigned)e, (unsigned)(e >> 32), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);        // (0, 1)
    const float u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1)) * 1e-5f;
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    n0 = rad * cs; n1 = rad * sn;
}

__global__ __launch_bounds__(256) void tie_break_noise_kernel(float* __restrict__ out, size_t npix, unsigned long long seed,
                                                              unsigned long long offset) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= npix) return;
    float a, b;
    tie_break_pair(seed, offset, e, a, b);
    out[2 * e] = a; out[2 * e + 1] = b;
}

template <bool TRAIN>
__global__ __launch_bounds__(256) void photo_automask_kernel(const float* __restrict__ warped_all, const float* __restrict__ target,
                                                             const float* __restrict__ idmap, const float* __restrict__ noise_all,
                                                             unsigned char* __restrict__ sel_all, float* __restrict__ coef_sel_all,
                                                             float* __restrict__ partial_all, int B, int H, int W, int tilesX,
                                                             unsigned long long seed, unsigned long long rng_offset, int sc0) {
    // The two source frames of a pixel travel as ONE packed pair (f32x2 = v_pk_* arithmetic, ds_read_b64): the kernel is
    // VALU-bound (~900 issue slots per pixel before, two thirds of them per frame), not LDS- or HBM-bound.
    __shared__ float tl[3][PA_PH * PA_PW];   // target
    __shared__ f32x2 tw[3][PA_PH * PA_PW];   // warped (frame 0, frame 1)
    __shared__ float red[4];
    const float C1 = 0.0001f, C2 = 0.0009f;
    const int b = blockIdx.y, sc = blockIdx.z + sc0;      // sc0: first scale of this launch
    const int HW = H * W;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int y0 = ty * PA_TH, x0 = tx * PA_TW;
    const float* warped = warped_all + (size_t)sc * 2 * B * 3 * HW;
    const float* noise = noise_all ? noise_all + (size_t)sc * B * 2 * HW : nullptr;
    unsigned char* sel = sel_all + ((size_t)sc * B + b) * HW;
    float* coef_sel = TRAIN ? coef_sel_all + ((size_t)sc * B + b) * 9 * HW : nullptr;
    const float* tgt = target + (size_t)b * 3 * HW;
    const float* wp0 = warped + ((size_t)0 * B + b) * 3 * HW;
    const float* wp1 = warped + ((size_t)1 * B + b) * 3 * HW;
    for (int e = threadIdx.x; e < PA_PH * PA_PW; e += 256) {
        const int r = e / PA_PW, c = e - r * PA_PW;
        // reflection padding; tiles overhanging the image clamp (those pixels are never consumed)
        const int yy = min(max(refl(y0 - 1 + r, H), 0), H - 1), xx = min(max(refl(x0 - 1 + c, W), 0), W - 1);
        const unsigned o = (unsigned)(yy * W + xx);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            tl[c3][e] = (tgt + (size_t)c3 * HW)[o];
            f32x2 v;
            v.x = (wp0 + (size_t)c3 * HW)[o];
            v.y = (wp1 + (size_t)c3 * HW)[o];
            tw[c3][e] = v;
        }
    }
    __syncthreads();
    float s = 0.f;
#pragma unroll 1
    for (int it = 0; it < (PA_TH * PA_TW) / 256; ++it) {
        const int lp = threadIdx.x + it * 256;
        const int ly = lp / PA_TW, lx = lp - ly * PA_TW;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        const unsigned p = (unsigned)(y * W + x);
        f32x2 cm = {0.f, 0.f}, l1 = {0.f, 0.f};   // per frame: sum_c clamp((1-SSIM)/2), sum_c |t - p|
        f32x2 kc[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float tv[9];
            float sy = 0.f, syy = 0.f;
            f32x2 sx = {0.f, 0.f}, sxx = {0.f, 0.f}, sxy = {0.f, 0.f}, xc = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int e = (ly + k / 3) * PA_PW + lx + k % 3;
                tv[k] = tl[c][e];
                sy += tv[k]; syy += tv[k] * tv[k];
                const f32x2 xv = tw[c][e];
                if (k == 4) xc = xv;
                sx += xv; sxx += xv * xv; sxy += xv * tv[k];
            }
            const float mu_y = div9(sy);
            const float sig_y = div9(syy) - mu_y * mu_y;
            const f32x2 mu_x = div9(sx);
            const f32x2 sig_x = div9(sxx) - mu_x * mu_x, sig_xy = div9(sxy) - mu_x * mu_y;
            const f32x2 n1 = 2.f * mu_x * mu_y + C1, n2 = 2.f * sig_xy + C2;
            const f32x2 d1 = mu_x * mu_x + mu_y * mu_y + C1, d2 = sig_x + sig_y + C2;
            const f32x2 dd = d1 * d2;
            f32x2 inv_d;
            inv_d.x = fast_rcp(dd.x); inv_d.y = fast_rcp(dd.y);
            const f32x2 S = (n1 * n2) * inv_d;
            const f32x2 raw = (1.f - S) * 0.5f;
            cm += pk_min(pk_max(raw, 0.f), 1.f);
            l1 += pk_abs(tv[4] - xc);
            if constexpr (TRAIN) {
                f32x2 kf;
                kf.x = (raw.x >= 0.f && raw.x <= 1.f) ? (0.85f / 3.f) * (-0.5f) / 9.f : 0.f;
                kf.y = (raw.y >= 0.f && raw.y <= 1.f) ? (0.85f / 3.f) * (-0.5f) / 9.f : 0.f;
                kc[c * 3 + 0] = kf * inv_d * (2.f * mu_y * (n2 - n1) - S * (2.f * mu_x * (d2 - d1)));
                kc[c * 3 + 1] = kf * inv_d * (-2.f * S * d1);
                kc[c * 3 + 2] = kf * inv_d * (2.f * n1);
            }
        }
        const float c2 = (0.85f / 3.f) * cm.x + (0.15f / 3.f) * l1.x;
        const float c3 = (0.85f / 3.f) * cm.y + (0.15f / 3.f) * l1.y;
        float c0 = idmap[((size_t)0 * B + b) * HW + p];
        float c1 = idmap[((size_t)1 * B + b) * HW + p];
        if (noise) { c0 += noise[((size_t)b * 2 + 0) * HW + p]; c1 += noise[((size_t)b * 2 + 1) * HW + p]; }
        else if (seed) {
            float z0, z1;
            tie_break_pair(seed, rng_offset, ((size_t)sc * B + b) * HW + p, z0, z1);
            c0 += z0; c1 += z1;
        }
        float m = c0; int k = 0;
        if (c1 < m) { m = c1; k = 1; }
        if (c2 < m) { m = c2; k = 2; }
        if (c3 < m) { m = c3; k = 3; }
        sel[p] = (unsigned char)k;
        if constexpr (TRAIN) {
            if (k >= 2) {
#pragma unroll
                for (int q = 0; q < 9; ++q) (coef_sel + (size_t)q * HW)[p] = (k == 2) ? kc[q].x : kc[q].y;
            }
        }
        s += m;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial_all[((size_t)sc * B + b) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Loss backward v2 for the whole pyramid, LDS-tiled: a block owns an 8 x 64 pixel tile of one (scale,
// sample); sel and the selected-frame coefficients of the tile + 1-pixel halo are staged once in LDS
// (every neighbour a pixel needs -- incl. the reflection fold -- lies in its 3x3 neighbourhood), then each
// thread does the transposed SSIM stencil from LDS and the grid_sample / projection / depth backward.
constexpr int LB_TH = 8, LB_TW = 64, LB_PH = LB_TH + 2, LB_PW = LB_TW + 2;
__global__ __launch_bounds__(256) void loss_bwd2_kernel(Pyramid pyr, const unsigned char* __restrict__ sel_all,
                                                        const float* __restrict__ coef_sel_all, const float* __restrict__ warped_all,
                                                        const float* __restrict__ target, const float* __restrict__ src_m1,
                                                        const float* __restrict__ src_p1, const float* __restrict__ Kinv,
                                                        const float* __restrict__ P, const float* __restrict__ sample_w,
                                                        float* __restrict__ ddisp_up_all, double* __restrict__ dP_partial, int B,
                                                        int H, int W, float da, float db, int dmode, int tilesX) {
    // coefficients pixel-major, 10 floats per pixel as five pairs (alpha_0 alpha_1)(alpha_2 beta_0)(beta_1 beta_2)
    // (gamma_0 gamma_1)(gamma_2 -): a neighbour is five ds_read_b64 + five v_pk_fma instead of nine reads + nine FMAs
    // (40-byte stride: conflict-free for 8-byte reads)
    __shared__ f32x2 cs[LB_PH * LB_PW][5];
    __shared__ unsigned char ss[LB_PH * LB_PW];
    __shared__ double red[4][24];     // the pose-gradient sums leave the thread in double (wave_sum_f64)
    const int b = blockIdx.y, sc = blockIdx.z + pyr.base;
    const int HW = H * W;
    const int ty = blockIdx.x / tilesX, tx = blockIdx.x - ty * tilesX;
    const int y0 = ty * LB_TH, x0 = tx * LB_TW;
    const int h = pyr.h[sc], w = pyr.w[sc];
    const float* disp_s = pyr.disp[sc];
    const unsigned char* sl = sel_all + ((size_t)sc * B + b) * HW;
    const float* coef = coef_sel_all + ((size_t)sc * B + b) * 9 * HW;
    const float* warped = warped_all + (size_t)sc * 2 * B * 3 * HW;
    float* ddisp_up = ddisp_up_all + (size_t)sc * B * HW;
    const float wq = sample_w[b] / ((float)H * (float)W) / 4.f;
    // ---- stage sel + coefficients (tile + halo; outside the image: sel = 255 -> never matches) -------------
    for (int e = threadIdx.x; e < LB_PH * LB_PW; e += 256) {
        const int r = e / LB_PW, c = e - r * LB_PW;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const unsigned char sv = in ? sl[yy * W + xx] : (unsigned char)255;
        ss[e] = sv;
        const bool need = in && sv >= 2 && sv < 4;
        const unsigned o = need ? (unsigned)(yy * W + xx) : 0u;
        float v[10];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) { const float t = (coef + (size_t)(c * 3 + j) * HW)[o]; v[j * 3 + c] = need ? t : 0.f; }
        v[9] = 0.f;
#pragma unroll
        for (int q = 0; q < 5; ++q) { f32x2 t; t.x = v[2 * q]; t.y = v[2 * q + 1]; cs[e][q] = t; }
    }
    __syncthreads();
    const float* Ki = Kinv + (size_t)b * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // One pass per source frame: only 12 pose-gradient accumulators and one set of stencil sums are live at
    // a time (the two-frame version needed 152 VGPRs = 3 waves per SIMD).
#pragma unroll 1
    for (int fi = 0; fi < 2; ++fi) {
        const int n = fi * B + b;
        const float* Pm = P + (size_t)n * 12;
        const float* src = (fi == 0 ? src_m1 : src_p1) + (size_t)b * 3 * HW;
        const unsigned char want = (unsigned char)(2 + fi);
        float dPacc[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) dPacc[k] = 0.f;
#pragma unroll 1
        for (int it = 0; it < (LB_TH * LB_TW) / 256; ++it) {
            const int lp = threadIdx.x + it * 256;
            const int ly = lp / LB_TW, lx = lp - ly * LB_TW;
            const int y = y0 + ly, x = x0 + lx;
            if (y >= H || x >= W) continue;
            const int pi = y * W + x;
            // transposed stencil: weights of the 3x3 neighbours incl. the reflection fold (rows/cols 0 and H-1 /
            // W-1 are counted twice for the second / second-to-last row / column); branch-free: neighbours that
            // selected the other frame (or none) enter with weight 0
            float wy[3] = {1.f, 1.f, 1.f}, wx[3] = {1.f, 1.f, 1.f};
            if (y == 1) wy[0] = 2.f;
            if (y == H - 2) wy[2] = 2.f;
            if (x == 1) wx[0] = 2.f;
            if (x == W - 2) wx[2] = 2.f;
            // Nothing to do for this frame where no pixel of the 3x3 neighbourhood selected it (the automask / the other
            // frame won): the selection is spatially coherent, so whole waves (64 consecutive pixels of a row) skip the
            // 81 coefficient reads, the projection and the gathers
            bool any = false;
#pragma unroll
            for (int q = 0; q < 9; ++q) any |= ss[(ly + q / 3) * LB_PW + lx + q % 3] == want;
            if (!any) {
                if (fi == 0) ddisp_up[(size_t)b * HW + pi] = 0.f;
                continue;
            }
            f32x2 acc[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) { acc[q].x = 0.f; acc[q].y = 0.f; }
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int e = (ly + dy) * LB_PW + lx + dx;
                    const float w1 = ss[e] == want ? (dy == 0 ? wy[0] : dy == 1 ? wy[1] : wy[2]) * wx[dx] : 0.f;
                    const f32x2 ww = {w1, w1};
#pragma unroll
                    for (int q = 0; q < 5; ++q) acc[q] = pk_fma(ww, cs[e][q], acc[q]);
                }
            const float sA[3] = {acc[0].x, acc[0].y, acc[1].x}, sB[3] = {acc[1].y, acc[2].x, acc[2].y},
                        sC[3] = {acc[3].x, acc[3].y, acc[4].x};
            const bool own = ss[(ly + 1) * LB_PW + lx + 1] == want;
            const float disp = upsample_disp(disp_s + (size_t)b * h * w, h, w, H, W, y, x);
            const float dep = disp_to_depth_dev(disp, da, db, dmode);
            const float fx = (float)x, fy = (float)y;
            float cam[3], X[3];
            backproject_px(Ki, fx, fy, dep, cam, X);
            float u, v, den;
            project_px(Pm, X, u, v, den);      // bit for bit the forward's position (geometry_dev.h)
            const Sample s = sample_coords(u, v, H, W);
            const float wx1 = s.ix - (float)s.x0, wy1 = s.iy - (float)s.y0;
            const float wx0 = (float)(s.x0 + 1) - s.ix, wy0 = (float)(s.y0 + 1) - s.iy;
            const bool x1ok = s.x0 + 1 < W, y1ok = s.y0 + 1 < H;
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float xv = warped[((size_t)n * 3 + c) * HW + pi];
                const float yv = target[((size_t)b * 3 + c) * HW + pi];
                float g = sA[c] + sB[c] * xv + sC[c] * yv;
                if (own) g += (0.15f / 3.f) * (xv > yv ? 1.f : (xv < yv ? -1.f : 0.f));
                g *= wq;
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
            const float inv_den = fast_rcp(den);   // u, v above keep the exact quotient: they pick the bilinear cell
            dp[0] = du * inv_den; dp[1] = dv * inv_den; dp[2] = -(du * u + dv * v) * inv_den;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                dPacc[i * 4 + 0] += dp[i] * X[0]; dPacc[i * 4 + 1] += dp[i] * X[1];
                dPacc[i * 4 + 2] += dp[i] * X[2]; dPacc[i * 4 + 3] += dp[i];
            }
            const float ddepth = ddepth_from_duv(Pm, cam, du, dv, inv_den);
            const float dd = (dmode == 2) ? -db * dep * dep * ddepth : -dep / disp * ddepth;
            // frame 0 writes, frame 1 adds (same thread, same address: ordered)
            float* o = ddisp_up + (size_t)b * HW + pi;
            *o = fi == 0 ? dd : *o + dd;
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const double s = wave_sum_f64((double)dPacc[k]);     // a thread holds two pixels' terms; from here on: double
            if (lane == 0) red[wave][fi * 12 + k] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < 24)
        dP_partial[(((size_t)sc * B + b) * gridDim.x + blockIdx.x) * 24 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// map[n,y,x] = 0.85*mean_c ssim_c + 0.15*mean_c |t-p|  for pred image n (N = npred images, each
// (3,H,W)) against target image (n % B).  When coef != NULL also stores, per channel c, the three
// coefficients (alpha, beta, gamma) of d map / d pred_window_element = alpha + beta*x_r + gamma*y_r
// (SSIM part only; the L1 part is recomputed in photo_grad) as coef[n][c*3+{0,1,2}][y][x].
__global__ __launch_bounds__(256) void photo_map_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        float* __restrict__ map, float* __restrict__ coef, int N, int B,
                                                        int H, int W) {
    const float C1 = 0.0001f, C2 = 0.0009f;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)N * HW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / HW);
        const int b = n % B;
        int ys[3], xs[3];
        for (int k = 0; k < 3; ++k) { ys[k] = refl(y + k - 1, H); xs[k] = refl(x + k - 1, W); }
        float ssim_sum = 0.f, l1_sum = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float* pp = pred + ((size_t)n * 3 + c) * HW;
            const float* tp = target + ((size_t)b * 3 + c) * HW;
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const float xv = pp[ys[ky] * W + xs[kx]], yv = tp[ys[ky] * W + xs[kx]];
                    sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                }
            const float mu_x = sx / 9.f, mu_y = sy / 9.f;
            const float sig_x = sxx / 9.f - mu_x * mu_x, sig_y = syy / 9.f - mu_y * mu_y, sig_xy = sxy / 9.f - mu_x * mu_y;
            const float n1 = 2.f * mu_x * mu_y + C1, n2 = 2.f * sig_xy + C2;
            const float d1 = mu_x * mu_x + mu_y * mu_y + C1, d2 = sig_x + sig_y + C2;
            const float d = d1 * d2;
            const float S = (n1 * n2) / d;
            const float raw = (1.f - S) / 2.f;
            ssim_sum += fminf(fmaxf(raw, 0.f), 1.f);
            l1_sum += fabsf(tp[y * W + x] - pp[y * W + x]);
            if (coef) {
                // d clamp((1-S)/2)/dS = -1/2 inside [0,1]; mean over 3 channels; 0.85 weight; /9 window
                const float kf = (raw >= 0.f && raw <= 1.f) ? (0.85f / 3.f) * (-0.5f) / 9.f : 0.f;
                const float A = (2.f * mu_y * (n2 - n1)) / d - S * (2.f * mu_x * (d2 - d1)) / d;
                const float Bc = -2.f * S * d1 / d;
                const float Cc = 2.f * n1 / d;
                float* co = coef + ((size_t)n * 9 + c * 3) * HW + (size_t)y * W + x;
                co[0] = kf * A; co[HW] = kf * Bc; co[2 * HW] = kf * Cc;
            }
        }
        map[idx] = 0.85f * (ssim_sum / 3.f) + 0.15f * (l1_sum / 3.f);
    }
}

// ------------------------------------------------------------------------------------------------
// to_optimize = min over [id(-1)+noise0, id(+1)+noise1, reproj(-1), reproj(+1)]; sel = argmin;
// partial[b][blk] = block sum of to_optimize.  grid (nblk, B).
__global__ __launch_bounds__(256) void automask_kernel(const float* __restrict__ idmap, const float* noise,
                                                       const float* rpmap, unsigned char* sel, float* partial, int B,
                                                       int HW, int pix_per_block) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    {   // blockIdx.z = scale of the pyramid: per-scale slices of noise / rpmap / sel / partial
        const size_t sc = blockIdx.z;
        if (noise) noise += sc * (size_t)B * 2 * HW;
        rpmap += sc * (size_t)2 * B * HW;
        sel += sc * (size_t)B * HW;
        partial += sc * (size_t)B * gridDim.x;
    }
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    float s = 0.f;
    for (int p = p0 + (int)threadIdx.x; p < p1; p += 256) {
        float c0 = idmap[((size_t)0 * B + b) * HW + p];
        float c1 = idmap[((size_t)1 * B + b) * HW + p];
        if (noise) { c0 += noise[((size_t)b * 2 + 0) * HW + p]; c1 += noise[((size_t)b * 2 + 1) * HW + p]; }
        const float c2 = rpmap[((size_t)0 * B + b) * HW + p];
        const float c3 = rpmap[((size_t)1 * B + b) * HW + p];
        float m = c0; int k = 0;
        if (c1 < m) { m = c1; k = 1; }
        if (c2 < m) { m = c2; k = 2; }
        if (c3 < m) { m = c3; k = 3; }
        sel[(size_t)b * HW + p] = (unsigned char)k;
        s += m;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// per-sample disparity sums in DM_CHUNKS pieces: psum[b][chunk] (grid (DM_CHUNKS, B)); the finalize
// kernel adds the chunks in order and divides by h*w.
constexpr int DM_CHUNKS = 32;
__global__ __launch_bounds__(256) void disp_mean_kernel(Pyramid pyr, float* psum, int B) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int sc = blockIdx.z;
    const float* disp = pyr.disp[sc];
    const int hw = pyr.h[sc] * pyr.w[sc];
    psum += (size_t)sc * B * DM_CHUNKS;
    const int per = (hw + DM_CHUNKS - 1) / DM_CHUNKS;
    const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
    float s = 0.f;
    for (int p = p0 + (int)threadIdx.x; p < p1; p += 256) s += disp[(size_t)b * hw + p];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) psum[(size_t)b * DM_CHUNKS + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

struct FinalizeArgs {
    const float* partial[4];     // [B][nblk] per scale
    const float* disp[4];        // (B,h_s,w_s)
    const float* rgb0[4];        // target pyramid ('rgb',0,s): (B,3,h_s,w_s)
    const float* means[4];       // per-sample disparity chunk sums [B][DM_CHUNKS]
    const float* pose;           // (2B,12)
    const double* dist0;         // |relative_distance(0)| source, (B)
    const double* dist1;
    const float* sample_w;       // (B) weights of the local samples
    const float* smooth_w;       // (n_smooth) weights of the smoothness terms (reference quirk)
    float* losses;               // [18]
    float* smooth_aux;           // [4][2 + 2*n_smooth]: inv, dsum, gxs[i], gys[i]
    int B, nblk, H, W, n_smooth;
    float smooth_scale, vel_scale;
};

// One block of 256 threads.  Phase 1 spreads the O(B*nblk) partial sums, the O(B) smoothness terms
// and the O(B) velocity terms over the lanes; phase 2 (thread 0) combines them in a fixed order.
constexpr int FIN_MAXB = 64;
__global__ __launch_bounds__(256) void loss_finalize_kernel(FinalizeArgs a) {
    __shared__ float s_rl[4 * FIN_MAXB];      // weighted per-sample reprojection means
    __shared__ float s_sm[4 * FIN_MAXB];      // weighted smoothness terms
    __shared__ float s_ds[4 * FIN_MAXB];      // per-term contributions to sum_q D[q]*disp[q]
    __shared__ float s_vel[FIN_MAXB];
    __shared__ float s_mean[4 * FIN_MAXB];
    const int tid = threadIdx.x;
    const float invHW = 1.f / ((float)a.H * (float)a.W);
    for (int pr = tid; pr < 4 * a.B; pr += 256) {
        const int s = pr / a.B, b = pr - s * a.B;
        float sum = 0.f;
        for (int k = 0; k < DM_CHUNKS; ++k) sum += a.means[s][(size_t)b * DM_CHUNKS + k];
        s_mean[pr] = sum / (float)((a.H >> s) * (a.W >> s));
    }
    __syncthreads();
    // 8 lanes per (scale, sample) pair: nblk is one partial per 8 x 64 tile (240 at 192 x 640), a serial chain
    // of that many dependent-latency loads per thread otherwise
    for (int pr0 = 0; pr0 < 4 * a.B; pr0 += 32) {
        const int pr = pr0 + (tid >> 3), kl = tid & 7;
        const bool live = pr < 4 * a.B;
        const int s = live ? pr / a.B : 0, b = live ? pr - s * a.B : 0;
        float sum = 0.f;
        if (live)
            for (int k = kl; k < a.nblk; k += 8) sum += a.partial[s][(size_t)b * a.nblk + k];
        sum += wave_shfl_xor(sum, 1);
        sum += wave_shfl_xor(sum, 2);
        sum += wave_shfl_xor(sum, 4);
        if (live && kl == 0) s_rl[pr] = (sum * invHW) * a.sample_w[b];
    }
    // smoothness, reference quirk: term i = flat element i of the batch-flattened gradient maps
    for (int pr = tid; pr < 4 * a.n_smooth; pr += 256) {
        const int s = pr / a.n_smooth, i = pr - s * a.n_smooth;
        const int h = a.H >> s, w = a.W >> s;
        float* aux = a.smooth_aux + (size_t)s * (2 + 2 * a.n_smooth);
        const float coefs = a.smooth_scale / (float)(1 << s) / 4.f;
        // flat index i in (B,1,h,w-1) and (B,1,h-1,w)
        const int bx = i / (h * (w - 1)), rx = i % (h * (w - 1)), yx = rx / (w - 1), xx = rx % (w - 1);
        const int by = i / ((h - 1) * w), ry = i % ((h - 1) * w), yy = ry / w, xy = ry % w;
        const float* dxp = a.disp[s] + (size_t)bx * h * w;
        const float* dyp = a.disp[s] + (size_t)by * h * w;
        const float mx = s_mean[s * a.B + bx] + 1e-7f, my = s_mean[s * a.B + by] + 1e-7f;
        const float ax = dxp[yx * w + xx] / mx, bxv = dxp[yx * w + xx + 1] / mx;   // norm_disp (dpp.py:1087-1088)
        const float ay = dyp[yy * w + xy] / my, byv = dyp[(yy + 1) * w + xy] / my;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float* im_x = a.rgb0[s] + ((size_t)bx * 3 + c) * h * w;
            const float* im_y = a.rgb0[s] + ((size_t)by * 3 + c) * h * w;
            gix += fabsf(im_x[yx * w + xx] - im_x[yx * w + xx + 1]);
            giy += fabsf(im_y[yy * w + xy] - im_y[(yy + 1) * w + xy]);
        }
        const float ex = expf(-(gix / 3.f)), ey = expf(-(giy / 3.f));
        s_sm[pr] = (fabsf(ax - bxv) * ex + fabsf(ay - byv) * ey) * a.smooth_w[i];
        // d/d norm_disp bookkeeping for the backward (layout i < w-1 on sample 0 enforced by the host)
        const float sgx = (ax > bxv) ? 1.f : (ax < bxv ? -1.f : 0.f);
        const float sgy = (ay > byv) ? 1.f : (ay < byv ? -1.f : 0.f);
        const float gxs = coefs * a.smooth_w[i] * ex * sgx;
        const float gys = coefs * a.smooth_w[i] * ey * sgy;
        aux[2 + i] = gxs;
        aux[2 + a.n_smooth + i] = gys;
        s_ds[pr] = gxs * (dxp[yx * w + xx] - dxp[yx * w + xx + 1]) + gys * (dyp[yy * w + xy] - dyp[(yy + 1) * w + xy]);
    }
    if (a.vel_scale > 0.f) {
        for (int b = tid; b < a.B; b += 256) {
            float acc = 0.f;
            for (int fi = 0; fi < 2; ++fi) {
                const float* t = a.pose + ((size_t)fi * a.B + b) * 12 + 3;
                const float nrm = sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
                const double gt = fabs(fi == 0 ? a.dist0[b] : a.dist1[b]);
                acc = (float)((double)acc + fabs((double)nrm - gt));   // fp32 += fp64 (dpp.py:1142)
            }
            s_vel[b] = (a.vel_scale * (acc / 2.f)) * a.sample_w[b];
        }
    }
    __syncthreads();
    if (tid != 0) return;
    float total = 0.f;
    for (int s = 0; s < 4; ++s) {
        const int h = a.H >> s, w = a.W >> s;
        float rl = 0.f, sm = 0.f, dsum = 0.f;
        for (int b = 0; b < a.B; ++b) rl += s_rl[s * a.B + b];
        for (int i = 0; i < a.n_smooth; ++i) { sm += s_sm[s * a.n_smooth + i]; dsum += s_ds[s * a.n_smooth + i]; }
        if (a.n_smooth > 0) {
            float* aux = a.smooth_aux + (size_t)s * (2 + 2 * a.n_smooth);
            const float inv0 = 1.f / (s_mean[s * a.B + 0] + 1e-7f);
            aux[0] = inv0;
            aux[1] = dsum * inv0 * inv0 / (float)(h * w);   // mean-normalisation feedback term
        }
        const float reg = a.smooth_scale / (float)(1 << s) * sm;
        const float loss = rl + reg;
        a.losses[s * 4 + 0] = rl; a.losses[s * 4 + 1] = sm; a.losses[s * 4 + 2] = reg; a.losses[s * 4 + 3] = loss;
        total += loss;
    }
    total = total / 4.f;
    float vel = 0.f;
    if (a.vel_scale > 0.f) {
        for (int b = 0; b < a.B; ++b) vel += s_vel[b];
        total += vel;
    }
    a.losses[16] = vel;
    a.losses[17] = total;
}

// ------------------------------------------------------------------------------------------------
// dL/d warped[n = fi*B+b][c][y][x] (un-weighted) for one scale: transposed SSIM stencil over the pixels
// whose min picked reprojection fi (sel == 2+fi), reflection fold, plus the L1 term.
__device__ __forceinline__ void photo_grad_px(const unsigned char* __restrict__ sl, const float* __restrict__ coef_n,
                                              const float* __restrict__ pred_n, const float* __restrict__ target_b, int H,
                                              int W, int y, int x, unsigned char want, float g[3]) {
    const size_t HW = (size_t)H * W;
    int py[3], px[3], npy = 0, npx = 0;   // padded-domain positions that reflect onto (y,x)
    py[npy++] = y + 1; px[npx++] = x + 1;
    if (y == 1) py[npy++] = 0;
    if (y == H - 2) py[npy++] = H + 1;
    if (x == 1) px[npx++] = 0;
    if (x == W - 2) px[npx++] = W + 1;
    float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f}, sc[3] = {0.f, 0.f, 0.f};
    for (int iy = 0; iy < npy; ++iy)
        for (int ix = 0; ix < npx; ++ix) {
            const int qy0 = max(0, py[iy] - 2), qy1 = min(H - 1, py[iy]);
            const int qx0 = max(0, px[ix] - 2), qx1 = min(W - 1, px[ix]);
            for (int qy = qy0; qy <= qy1; ++qy)
                for (int qx = qx0; qx <= qx1; ++qx) {
                    if (sl[qy * W + qx] != want) continue;
                    const float* co = coef_n + (size_t)qy * W + qx;
                    for (int c = 0; c < 3; ++c) {
                        sa[c] += co[(c * 3 + 0) * HW];
                        sb[c] += co[(c * 3 + 1) * HW];
                        sc[c] += co[(c * 3 + 2) * HW];
                    }
                }
        }
    const bool own = sl[y * W + x] == want;
    for (int c = 0; c < 3; ++c) {
        const float xv = pred_n[(size_t)c * HW + (size_t)y * W + x];
        const float yv = target_b[(size_t)c * HW + (size_t)y * W + x];
        float v = sa[c] + sb[c] * xv + sc[c] * yv;
        if (own) v += (0.15f / 3.f) * (xv > yv ? 1.f : (xv < yv ? -1.f : 0.f));
        g[c] = v;
    }
}

// dpred[fi,b,c,y,x] = dL/d warped for one scale (stand-alone form, used by the kernel-level tests)
__global__ __launch_bounds__(256) void photo_grad_kernel(const unsigned char* __restrict__ sel, const float* __restrict__ coef,
                                                         const float* __restrict__ pred, const float* __restrict__ target,
                                                         const float* __restrict__ sample_w, float* __restrict__ dpred,
                                                         int B, int H, int W) {
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)2 * B * HW;
    const float scale_all = 1.f / ((float)H * (float)W) / 4.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W), y = (int)((idx / W) % H);
        const int n = (int)(idx / HW);  // fi*B + b
        const int fi = n / B, b = n - fi * B;
        float g[3];
        photo_grad_px(sel + (size_t)b * HW, coef + (size_t)n * 9 * HW, pred + (size_t)n * 3 * HW, target + (size_t)b * 3 * HW, H, W,
                      y, x, (unsigned char)(2 + fi), g);
        const float wq = sample_w[b] * scale_all;
        for (int c = 0; c < 3; ++c) dpred[((size_t)n * 3 + c) * HW + (size_t)y * W + x] = g[c] * wq;
    }
}

// Fused loss backward for the whole pyramid (one launch): photometric gradient -> grid_sample / projection
// / depth backward.  grid (nblk, B, nscale).  Writes ddisp_up[s,b,y,x] and dP_partial[s][b][blk][24].
__global__ __launch_bounds__(256) void loss_bwd_kernel(Pyramid pyr, const unsigned char* __restrict__ sel_all,
                                                       const float* __restrict__ coef_all, const float* __restrict__ warped_all,
                                                       const float* __restrict__ target, const float* __restrict__ src_m1,
                                                       const float* __restrict__ src_p1, const float* __restrict__ Kinv,
                                                       const float* __restrict__ P, const float* __restrict__ sample_w,
                                                       float* __restrict__ ddisp_up_all, double* __restrict__ dP_partial, int B,
                                                       int H, int W, float da, float db, int dmode, int pix_per_block) {
    __shared__ double red[4][24];
    const int b = blockIdx.y, sc = blockIdx.z;
    const int HW = H * W;
    const int h = pyr.h[sc], w = pyr.w[sc];
    const float* disp_s = pyr.disp[sc];
    const unsigned char* sl = sel_all + ((size_t)sc * B + b) * HW;
    const float* coef = coef_all + (size_t)sc * 2 * B * 9 * HW;
    const float* warped = warped_all + (size_t)sc * 2 * B * 3 * HW;
    float* ddisp_up = ddisp_up_all + (size_t)sc * B * HW;
    const float wq = sample_w[b] / ((float)H * (float)W) / 4.f;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    double dPacc[24];
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
#pragma unroll 1
        for (int fi = 0; fi < 2; ++fi) {   // not unrolled: halves the live registers (occupancy 2 -> 4 waves/SIMD)
            const int n = fi * B + b;
            float g3[3];
            photo_grad_px(sl, coef + (size_t)n * 9 * HW, warped + (size_t)n * 3 * HW, target + (size_t)b * 3 * HW, H, W, y, x,
                          (unsigned char)(2 + fi), g3);
            const float* Pm = P + (size_t)n * 12;
            float u, v, den;
            project_px(Pm, X, u, v, den);
            const Sample s = sample_coords(u, v, H, W);
            const float wx1 = s.ix - (float)s.x0, wy1 = s.iy - (float)s.y0;
            const float wx0 = (float)(s.x0 + 1) - s.ix, wy0 = (float)(s.y0 + 1) - s.iy;
            const bool x1ok = s.x0 + 1 < W, y1ok = s.y0 + 1 < H;
            const float* src = (fi == 0 ? src_m1 : src_p1) + (size_t)b * 3 * HW;
            float gix = 0.f, giy = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float g = g3[c] * wq;
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
            // dPacc is indexed with compile-time constants only (a runtime index would spill it to scratch)
            if (fi == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    dPacc[i * 4 + 0] += dp[i] * X[0]; dPacc[i * 4 + 1] += dp[i] * X[1];
                    dPacc[i * 4 + 2] += dp[i] * X[2]; dPacc[i * 4 + 3] += dp[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    dPacc[12 + i * 4 + 0] += dp[i] * X[0]; dPacc[12 + i * 4 + 1] += dp[i] * X[1];
                    dPacc[12 + i * 4 + 2] += dp[i] * X[2]; dPacc[12 + i * 4 + 3] += dp[i];
                }
            }
            ddepth += ddepth_from_duv(Pm, cam, du, dv, 1.f / den);
        }
        ddisp_up[(size_t)b * HW + pi] = (dmode == 2) ? -db * dep * dep * ddepth : -dep / disp * ddepth;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const double s = wave_sum_f64(dPacc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 24)
        dP_partial[(((size_t)sc * B + b) * gridDim.x + blockIdx.x) * 24 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// dz[b,i,j] = sigmoid'(disp) * ( bilinear-upsample^T(ddisp_up)[b,i,j] + smoothness gradient )
struct DzPtrs { float* dz[4]; };

__global__ __launch_bounds__(256) void disp_grad_kernel(const float* ddisp_up, const float* disp, const float* smooth_aux,
                                                        int n_smooth, float* dz, int B, int h, int w, int H, int W, Pyramid pyr,
                                                        DzPtrs dzp) {
    if (pyr.n > 0) {   // pyramid form: blockIdx.y = scale
        const int sc = blockIdx.y;
        h = pyr.h[sc]; w = pyr.w[sc];
        disp = pyr.disp[sc];
        dz = dzp.dz[sc];
        ddisp_up += (size_t)sc * B * H * W;
        if (smooth_aux) smooth_aux += (size_t)sc * (2 + 2 * n_smooth);
    }
    const size_t total = (size_t)B * h * w;
    const int f = H / h;
    const float ry = (float)h / (float)H, rx = (float)w / (float)W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % w), i = (int)((idx / w) % h), b = (int)(idx / ((size_t)w * h));
        float g = 0.f;
        const int Y0 = max(0, f * i - f), Y1 = min(H, f * i + 2 * f);
        const int X0 = max(0, f * j - f), X1 = min(W, f * j + 2 * f);
        for (int Y = Y0; Y < Y1; ++Y) {
            float sy = ry * ((float)Y + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
            const int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly = sy - (float)y0;
            float wy = 0.f;
            if (y0 == i) wy += 1.f - ly;
            if (y1 == i) wy += ly;
            if (wy == 0.f) continue;
            for (int X = X0; X < X1; ++X) {
                float sx = rx * ((float)X + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
                const int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx = sx - (float)x0;
                float wx = 0.f;
                if (x0 == j) wx += 1.f - lx;
                if (x1 == j) wx += lx;
                if (wx != 0.f) g += wy * wx * ddisp_up[((size_t)b * H + Y) * W + X];
            }
        }
        if (n_smooth > 0 && b == 0) {
            const float inv0 = smooth_aux[0], fb = smooth_aux[1];
            const float* gxs = smooth_aux + 2;
            const float* gys = smooth_aux + 2 + n_smooth;
            float D = 0.f;
            if (i == 0) {
                if (j < n_smooth) D += gxs[j] + gys[j];
                if (j >= 1 && j - 1 < n_smooth) D -= gxs[j - 1];
            } else if (i == 1) {
                if (j < n_smooth) D -= gys[j];
            }
            g += inv0 * D - fb;
        }
        const float d = disp[idx];
        dz[idx] = g * d * (1.f - d);
    }
}

// Pyramid form of disp_grad_kernel, lane-cooperative.  The bilinear footprint of low-res pixel (i,j) at
// factor f = H/h in {1,2,4,8} is the 2f x 2f window starting at (f*i - f/2, f*j - f/2); a one-thread-per-
// output loop over it is a long serial chain of dependent loads at the coarse scales (measured 96 us).
// Here f*f lanes share one output pixel (4 taps each: (qy, qx) + {0,f} in both axes; consecutive lanes
// read consecutive X), then xor-shuffle reduce: every scale exposes B*H*W threads of equal work.
__global__ __launch_bounds__(256) void disp_grad_coop_kernel(const float* __restrict__ ddisp_up_all, const float* __restrict__ smooth_all,
                                                             int n_smooth, int B, int H, int W, Pyramid pyr, DzPtrs dzp) {
    const int sc = blockIdx.y + pyr.base;
    const int h = pyr.h[sc], w = pyr.w[sc];
    const int f = H / h, G = f * f;
    const float* __restrict__ disp = pyr.disp[sc];
    float* __restrict__ dz = dzp.dz[sc];
    const float* __restrict__ ddisp_up = ddisp_up_all + (size_t)sc * B * H * W;
    const float* smooth_aux = smooth_all ? smooth_all + (size_t)sc * (2 + 2 * n_smooth) : nullptr;
    const float ry = (float)h / (float)H, rx = (float)w / (float)W;
    const int total = B * H * W;                       // = outputs * G
    const int t = blockIdx.x * 256 + threadIdx.x;      // grid covers total exactly up to the last block
    const bool live = t < total;
    const int idx = live ? t / G : 0, q = t % G;
    const int qy = q / f, qx = q % f;
    const int j = idx % w, i = (idx / w) % h, b = idx / (w * h);
    // The four taps and the disparity are requested together from clamped (always valid) addresses and weighted afterwards, in
    // the order of the loop they replace (a tap outside the image contributes an exact 0): with `continue` per tap every load
    // sat behind an `s_waitcnt vmcnt(0)` of its own -- five serial round trips on the step's critical chain (round 5).
    float g = 0.f;
    float wgt[4], val[4];
    const float dsp = disp[idx];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int Y = f * i - f / 2 + qy + a * f;
        const bool oky = Y >= 0 && Y < H;
        const int Yc = min(max(Y, 0), H - 1);
        float sy = ry * ((float)Yc + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
        const int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly = sy - (float)y0;
        float wy = 0.f;
        if (y0 == i) wy += 1.f - ly;
        if (y1 == i) wy += ly;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int X = f * j - f / 2 + qx + c * f;
            const bool okx = X >= 0 && X < W;
            const int Xc = min(max(X, 0), W - 1);
            float sx = rx * ((float)Xc + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            const int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float lx = sx - (float)x0;
            float wx = 0.f;
            if (x0 == j) wx += 1.f - lx;
            if (x1 == j) wx += lx;
            wgt[a * 2 + c] = (live && oky && okx) ? wy * wx : 0.f;
            val[a * 2 + c] = ddisp_up[((size_t)b * H + Yc) * W + Xc];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (wgt[k] != 0.f) g += wgt[k] * val[k];      // (a select, not a branch: same sum as the taps that were visited)
    for (int m = 1; m < G; m <<= 1) g += wave_shfl_xor(g, m);   // G <= 64 lanes of one wave, aligned
    if (live && q == 0) {
        if (n_smooth > 0 && b == 0) {
            const float inv0 = smooth_aux[0], fb = smooth_aux[1];
            const float* gxs = smooth_aux + 2;
            const float* gys = smooth_aux + 2 + n_smooth;
            float D = 0.f;
            if (i == 0) {
                if (j < n_smooth) D += gxs[j] + gys[j];
                if (j >= 1 && j - 1 < n_smooth) D -= gxs[j - 1];
            } else if (i == 1) {
                if (j < n_smooth) D -= gys[j];
            }
            g += inv0 * D - fb;
        }
        const float d = dsp;
        dz[idx] = g * d * (1.f - d);
    }
}

}  // namespace clslam

using namespace clslam;

static unsigned grid1d(size_t total) { return (unsigned)std::min<size_t>(8192, (total + 255) / 256); }

extern "C" int clslam_photo_map(const float* pred, const float* target, float* map, float* coef, int npred, int batch, int H,
                                int W, void* stream) {
    CLSLAM_REQUIRE(pred && target && map && batch > 0 && npred % batch == 0, "photo_map: bad args");
    CLSLAM_REQUIRE(H >= 2 && W >= 2, "photo_map: image too small for reflection padding");
    const size_t total = (size_t)npred * H * W;
    hipLaunchKernelGGL(photo_map_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, pred, target, map, coef, npred,
                       batch, H, W);
    return check_launch("photo_map");
}

// Partial sums per (scale, sample) written by the automask kernels = number of 8 x 64 pixel tiles.
extern "C" int clslam_automask_blocks(int H, int W) { return cdiv(H, PA_TH) * cdiv(W, PA_TW); }

extern "C" int clslam_automask(const float* idmap, const float* noise, const float* rpmap, unsigned char* sel, float* partial,
                               int batch, int H, int W, void* stream) {
    CLSLAM_REQUIRE(idmap && rpmap && sel && partial, "automask: null");
    if (!batch) return CLSLAM_OK;
    const int nblk = clslam_automask_blocks(H, W);
    hipLaunchKernelGGL(automask_kernel, dim3(nblk, batch), dim3(256), 0, (hipStream_t)stream, idmap, noise, rpmap, sel, partial,
                       batch, H * W, cdiv(H * W, nblk));
    return check_launch("automask");
}

// nscale scales in one launch: noise (S,B,2,H,W) or NULL, rpmap (S,2,B,H,W), sel (S,B,H,W), partial (S,B,nblk)
extern "C" int clslam_automask_pyramid(const float* idmap, const float* noise, const float* rpmap, unsigned char* sel,
                                       float* partial, int nscale, int batch, int H, int W, void* stream) {
    CLSLAM_REQUIRE(idmap && rpmap && sel && partial && nscale >= 1, "automask_pyramid: bad args");
    if (!batch) return CLSLAM_OK;
    const int nblk = clslam_automask_blocks(H, W);
    hipLaunchKernelGGL(automask_kernel, dim3(nblk, batch, nscale), dim3(256), 0, (hipStream_t)stream, idmap, noise, rpmap, sel,
                       partial, batch, H * W, cdiv(H * W, nblk));
    return check_launch("automask_pyramid");
}

extern "C" int clslam_disp_mean_chunks(void) { return DM_CHUNKS; }

extern "C" int clslam_disp_mean(const float* disp, float* means, int batch, int hw, void* stream) {
    CLSLAM_REQUIRE(disp && means, "disp_mean: null");
    if (!batch) return CLSLAM_OK;
    Pyramid pyr;
    pyr.n = 1; pyr.base = 0; pyr.disp[0] = disp; pyr.h[0] = 1; pyr.w[0] = hw;
    for (int k = 1; k < 4; ++k) { pyr.disp[k] = nullptr; pyr.h[k] = pyr.w[k] = 0; }
    hipLaunchKernelGGL(disp_mean_kernel, dim3(DM_CHUNKS, batch, 1), dim3(256), 0, (hipStream_t)stream, pyr, means, batch);
    return check_launch("disp_mean");
}

static Pyramid make_pyramid(const float* const* disp, int H, int W) {
    Pyramid pyr;
    pyr.n = 4; pyr.base = 0;
    for (int k = 0; k < 4; ++k) { pyr.disp[k] = disp[k]; pyr.h[k] = H >> k; pyr.w[k] = W >> k; }
    return pyr;
}

// psum (4,B,chunks) for the four disparity maps disp[s] (B,H>>s,W>>s), one launch
extern "C" int clslam_disp_mean_pyramid(const float* const* disp, float* psum, int batch, int H, int W, void* stream) {
    CLSLAM_REQUIRE(disp && psum, "disp_mean_pyramid: null");
    if (!batch) return CLSLAM_OK;
    hipLaunchKernelGGL(disp_mean_kernel, dim3(DM_CHUNKS, batch, 4), dim3(256), 0, (hipStream_t)stream, make_pyramid(disp, H, W),
                       psum, batch);
    return check_launch("disp_mean_pyramid");
}

// Fused photometric map + automask over the pyramid: warped (4,2,B,3,H,W), idmap (2,B,H,W), noise (4,B,2,H,W)|NULL ->
// sel (4,B,H,W), coef_sel (4,B,9,H,W)|NULL (training only), partial (4,B,clslam_automask_blocks).
static int photo_automask_launch(const float* warped, const float* target, const float* idmap, const float* noise,
                                 unsigned char* sel, float* coef_sel, float* partial, int batch, int H, int W,
                                 unsigned long long seed, unsigned long long offset, int scale_lo, int scale_count, void* stream) {
    CLSLAM_REQUIRE(warped && target && idmap && sel && partial && H >= 2 && W >= 2, "photo_automask_pyramid: bad args");
    CLSLAM_REQUIRE(scale_lo >= 0 && scale_count >= 0 && scale_lo + scale_count <= 4, "photo_automask_pyramid: scales [%d, %d) outside the pyramid",
                   scale_lo, scale_lo + scale_count);
    if (!batch || !scale_count) return CLSLAM_OK;
    const int nblk = clslam_automask_blocks(H, W);   // = number of 8 x 64 tiles
    const int tilesX = cdiv(W, PA_TW);
    if (coef_sel)
        hipLaunchKernelGGL(photo_automask_kernel<true>, dim3(nblk, batch, scale_count), dim3(256), 0, (hipStream_t)stream, warped, target,
                           idmap, noise, sel, coef_sel, partial, batch, H, W, tilesX, seed, offset, scale_lo);
    else
        hipLaunchKernelGGL(photo_automask_kernel<false>, dim3(nblk, batch, scale_count), dim3(256), 0, (hipStream_t)stream, warped, target,
                           idmap, noise, sel, coef_sel, partial, batch, H, W, tilesX, seed, offset, scale_lo);
    return check_launch("photo_automask_pyramid");
}

extern "C" int clslam_photo_automask_pyramid(const float* warped, const float* target, const float* idmap, const float* noise,
                                             unsigned char* sel, float* coef_sel, float* partial, int batch, int H, int W,
                                             void* stream) {
    return photo_automask_launch(warped, target, idmap, noise, sel, coef_sel, partial, batch, H, W, 0ull, 0ull, 0, 4, stream);
}

extern "C" int clslam_photo_automask_pyramid_rng(const float* warped, const float* target, const float* idmap,
                                                 unsigned long long seed, unsigned long long offset, unsigned char* sel,
                                                 float* coef_sel, float* partial, int batch, int H, int W, void* stream) {
    CLSLAM_REQUIRE(seed != 0, "photo_automask_pyramid_rng: seed must be non-zero");
    return photo_automask_launch(warped, target, idmap, nullptr, sel, coef_sel, partial, batch, H, W, seed, offset, 0, 4, stream);
}

// scales [scale_lo, scale_lo + scale_count) only: noise injected (noise != NULL), drawn in the kernel (seed != 0) or absent
extern "C" int clslam_photo_automask_pyramid_range(const float* warped, const float* target, const float* idmap, const float* noise,
                                                   unsigned long long seed, unsigned long long offset, unsigned char* sel,
                                                   float* coef_sel, float* partial, int batch, int H, int W, int scale_lo,
                                                   int scale_count, void* stream) {
    CLSLAM_REQUIRE(!(noise && seed), "photo_automask_pyramid_range: injected noise and an in-kernel draw exclude each other");
    return photo_automask_launch(warped, target, idmap, noise, sel, coef_sel, partial, batch, H, W, seed, offset, scale_lo, scale_count,
                                 stream);
}

extern "C" int clslam_tie_break_noise(float* out, size_t npix, unsigned long long seed, unsigned long long offset, void* stream) {
    CLSLAM_REQUIRE(out || npix == 0, "tie_break_noise: null");
    if (!npix) return CLSLAM_OK;
    hipLaunchKernelGGL(tie_break_noise_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, npix,
                       seed, offset);
    return check_launch("tie_break_noise");
}

extern "C" int clslam_loss_bwd2_blocks(int H, int W) { return cdiv(H, LB_TH) * cdiv(W, LB_TW); }

// LDS-tiled fused loss backward on the selected-frame coefficients (clslam_photo_automask_pyramid);
// dp_partial [4][B][clslam_loss_bwd2_blocks][24].
extern "C" int clslam_loss_bwd2_pyramid_range(const float* const* disp, const unsigned char* sel, const float* coef_sel,
                                              const float* warped, const float* target, const float* src_m1, const float* src_p1,
                                              const float* inv_k, const float* proj, const float* sample_w, float* ddisp_up,
                                              double* dp_partial, int batch, int H, int W, float min_depth, float max_depth,
                                              int scale_lo, int scale_count, void* stream) {
    CLSLAM_REQUIRE(disp && sel && coef_sel && warped && target && src_m1 && src_p1 && inv_k && proj && sample_w && ddisp_up &&
                   dp_partial, "loss_bwd2_pyramid: null");
    CLSLAM_REQUIRE(scale_lo >= 0 && scale_count >= 0 && scale_lo + scale_count <= 4, "loss_bwd2_pyramid: scales [%d, %d) outside the pyramid",
                   scale_lo, scale_lo + scale_count);
    float a, b; int mode;
    depth_mode(min_depth, max_depth, &a, &b, &mode);
    if (!batch || !scale_count) return CLSLAM_OK;
    const int tilesX = cdiv(W, LB_TW);
    Pyramid pyr = make_pyramid(disp, H, W);
    pyr.base = scale_lo;
    hipLaunchKernelGGL(loss_bwd2_kernel, dim3(clslam_loss_bwd2_blocks(H, W), batch, scale_count), dim3(256), 0, (hipStream_t)stream,
                       pyr, sel, coef_sel, warped, target, src_m1, src_p1, inv_k, proj, sample_w, ddisp_up,
                       dp_partial, batch, H, W, a, b, mode, tilesX);
    return check_launch("loss_bwd2_pyramid");
}

extern "C" int clslam_loss_bwd2_pyramid(const float* const* disp, const unsigned char* sel, const float* coef_sel,
                                        const float* warped, const float* target, const float* src_m1, const float* src_p1,
                                        const float* inv_k, const float* proj, const float* sample_w, float* ddisp_up,
                                        double* dp_partial, int batch, int H, int W, float min_depth, float max_depth, void* stream) {
    return clslam_loss_bwd2_pyramid_range(disp, sel, coef_sel, warped, target, src_m1, src_p1, inv_k, proj, sample_w, ddisp_up, dp_partial,
                                          batch, H, W, min_depth, max_depth, 0, 4, stream);
}

extern "C" int clslam_loss_bwd_blocks(int H, int W) { return std::max(1, std::min(256, cdiv(H * W, 1024))); }

// Fused photometric + view-synthesis backward of all four scales (one launch).  sel (4,B,H,W),
// coef (4,2,B,9,H,W), warped (4,2,B,3,H,W); ddisp_up (4,B,H,W); dp_partial [4][B][nblk][24].
extern "C" int clslam_loss_bwd_pyramid(const float* const* disp, const unsigned char* sel, const float* coef, const float* warped,
                                       const float* target, const float* src_m1, const float* src_p1, const float* inv_k,
                                       const float* proj, const float* sample_w, float* ddisp_up, double* dp_partial, int batch,
                                       int H, int W, float min_depth, float max_depth, void* stream) {
    CLSLAM_REQUIRE(disp && sel && coef && warped && target && src_m1 && src_p1 && inv_k && proj && sample_w && ddisp_up &&
                   dp_partial, "loss_bwd_pyramid: null");
    float a, b; int mode;
    depth_mode(min_depth, max_depth, &a, &b, &mode);
    if (!batch) return CLSLAM_OK;
    const int nblk = clslam_loss_bwd_blocks(H, W);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(nblk, batch, 4), dim3(256), 0, (hipStream_t)stream, make_pyramid(disp, H, W), sel, coef,
                       warped, target, src_m1, src_p1, inv_k, proj, sample_w, ddisp_up, dp_partial, batch, H, W, a, b, mode,
                       cdiv(H * W, nblk));
    return check_launch("loss_bwd_pyramid");
}

extern "C" int clslam_loss_finalize(const clslam_loss_desc* d, void* stream) {
    CLSLAM_REQUIRE(d && d->losses && d->pose && d->sample_w, "loss_finalize: null");
    CLSLAM_REQUIRE(d->n_smooth == 0 || (d->smooth_w && d->smooth_aux), "loss_finalize: smooth buffers missing");
    FinalizeArgs a;
    for (int s = 0; s < 4; ++s) { a.partial[s] = d->partial[s]; a.disp[s] = d->disp[s]; a.rgb0[s] = d->rgb0[s]; a.means[s] = d->means[s]; }
    a.pose = d->pose; a.dist0 = d->dist0; a.dist1 = d->dist1; a.sample_w = d->sample_w; a.smooth_w = d->smooth_w;
    a.losses = d->losses; a.smooth_aux = d->smooth_aux; a.B = d->batch; a.nblk = d->nblk; a.H = d->H; a.W = d->W;
    a.n_smooth = d->n_smooth; a.smooth_scale = d->smooth_scale; a.vel_scale = d->vel_scale;
    CLSLAM_REQUIRE(d->batch <= FIN_MAXB && d->n_smooth <= FIN_MAXB, "loss_finalize: batch > %d", FIN_MAXB);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("loss_finalize");
}

extern "C" int clslam_photo_grad(const unsigned char* sel, const float* coef, const float* pred, const float* target,
                                 const float* sample_w, float* dpred, int batch, int H, int W, void* stream) {
    CLSLAM_REQUIRE(sel && coef && pred && target && sample_w && dpred, "photo_grad: null");
    const size_t total = (size_t)2 * batch * H * W;
    if (!total) return CLSLAM_OK;
    hipLaunchKernelGGL(photo_grad_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, sel, coef, pred, target,
                       sample_w, dpred, batch, H, W);
    return check_launch("photo_grad");
}

extern "C" int clslam_disp_grad(const float* ddisp_up, const float* disp, const float* smooth_aux, int n_smooth, float* dz,
                                int batch, int h, int w, int H, int W, void* stream) {
    CLSLAM_REQUIRE(ddisp_up && disp && dz && H % h == 0 && W % w == 0 && H / h == W / w, "disp_grad: bad args");
    CLSLAM_REQUIRE(n_smooth == 0 || (smooth_aux && n_smooth < w - 1 && h >= 2), "disp_grad: smoothness layout unsupported");
    const size_t total = (size_t)batch * h * w;
    if (!total) return CLSLAM_OK;
    Pyramid none; none.n = 0;
    for (int k = 0; k < 4; ++k) { none.disp[k] = nullptr; none.h[k] = none.w[k] = 0; }
    DzPtrs dzp; for (int k = 0; k < 4; ++k) dzp.dz[k] = nullptr;
    hipLaunchKernelGGL(disp_grad_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, ddisp_up, disp, smooth_aux,
                       n_smooth, dz, batch, h, w, H, W, none, dzp);
    return check_launch("disp_grad");
}

// all four scales in one launch: ddisp_up (4,B,H,W), smooth_aux [4][2+2*n_smooth], dz[s] (B,H>>s,W>>s)
extern "C" int clslam_disp_grad_pyramid_range(const float* ddisp_up, const float* const* disp, const float* smooth_aux, int n_smooth,
                                              float* const* dz, int batch, int H, int W, int scale_lo, int scale_count, void* stream) {
    CLSLAM_REQUIRE(ddisp_up && disp && dz, "disp_grad_pyramid: null");
    CLSLAM_REQUIRE(n_smooth == 0 || (smooth_aux && n_smooth < (W >> 3) - 1 && (H >> 3) >= 2), "disp_grad_pyramid: smoothness layout unsupported");
    CLSLAM_REQUIRE(scale_lo >= 0 && scale_count >= 0 && scale_lo + scale_count <= 4, "disp_grad_pyramid: scales [%d, %d) outside the pyramid",
                   scale_lo, scale_lo + scale_count);
    if (!batch || !scale_count) return CLSLAM_OK;
    DzPtrs dzp; for (int k = 0; k < 4; ++k) dzp.dz[k] = dz[k];
    CLSLAM_REQUIRE(H % 8 == 0 && W % 8 == 0 && (size_t)batch * H * W < ((size_t)1 << 31), "disp_grad_pyramid: H, W must be multiples of 8");
    Pyramid pyr = make_pyramid(disp, H, W);
    pyr.base = scale_lo;
    hipLaunchKernelGGL(disp_grad_coop_kernel, dim3(cdiv(batch * H * W, 256), scale_count), dim3(256), 0, (hipStream_t)stream, ddisp_up,
                       smooth_aux, n_smooth, batch, H, W, pyr, dzp);
    return check_launch("disp_grad_pyramid");
}

extern "C" int clslam_disp_grad_pyramid(const float* ddisp_up, const float* const* disp, const float* smooth_aux, int n_smooth,
                                        float* const* dz, int batch, int H, int W, void* stream) {
    return clslam_disp_grad_pyramid_range(ddisp_up, disp, smooth_aux, n_smooth, dz, batch, H, W, 0, 4, stream);
}


// ------------------------------------------------------------------------------------------------------------------
// Opt-in "intended" smoothness: the per-sample edge-aware term the reference evidently meant
// (monodepth2), instead of the flattened-batch behaviour of dpp.py:1148-1176 that parity reproduces by default:
//   sm_b = mean_{y, x<w-1} |n(y,x) - n(y,x+1)| exp(-mean_c |I(y,x) - I(y,x+1)|) + the same along y,   n = disp / (mean(disp) + 1e-7)
// n is disp times a positive per-sample scalar, so sm_b = inv_b * T_b with T_b the same sums over the raw disparity.
// Three small kernels around the existing stage: chunked partial sums of T (fixed order), a one-block finalize that adds
// the terms to the 18 loss scalars and keeps (inv_b, T_b) for the backward, and the gradient added into dz of the
// disparity heads:  d sm_b / d disp(p) = inv_b * sum_{edges at p} sign * e / norm  -  T_b * inv_b^2 / (h w).
namespace clslam {

constexpr int SI_CHUNKS = 32;

__device__ __forceinline__ float si_edge_weight(const float* __restrict__ img, int hw, int p, int q) {
    float g = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) g += fabsf(img[(size_t)c * hw + p] - img[(size_t)c * hw + q]);
    return expf(-(g / 3.f));
}

struct SiPtrs { const float* rgb0[4]; };
struct SiDz { float* dz[4]; };

__global__ __launch_bounds__(256) void smooth_intended_fwd_kernel(Pyramid pyr, SiPtrs im, float* __restrict__ partial, int B) {
    __shared__ float red[4];
    const int b = blockIdx.y, sc = blockIdx.z;
    const int h = pyr.h[sc], w = pyr.w[sc], hw = h * w;
    const float* d = pyr.disp[sc] + (size_t)b * hw;
    const float* img = im.rgb0[sc] + (size_t)b * 3 * hw;
    const float nx = 1.f / (float)(h * (w - 1)), ny = 1.f / (float)((h - 1) * w);
    const int per = (hw + SI_CHUNKS - 1) / SI_CHUNKS;
    const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
    float s = 0.f;
    for (int p = p0 + (int)threadIdx.x; p < p1; p += 256) {
        const int y = p / w, x = p - y * w;
        if (x + 1 < w) s += fabsf(d[p] - d[p + 1]) * si_edge_weight(img, hw, p, p + 1) * nx;
        if (y + 1 < h) s += fabsf(d[p] - d[p + w]) * si_edge_weight(img, hw, p, p + w) * ny;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[((size_t)sc * B + b) * SI_CHUNKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one thread: 4 x B terms, fixed order
__global__ void smooth_intended_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ means_all,
                                                const float* __restrict__ sample_w, float* __restrict__ losses,
                                                float* __restrict__ aux, int B, int H, int W, float smooth_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float add_total = 0.f;
    for (int s = 0; s < 4; ++s) {
        const int hw = (H >> s) * (W >> s);
        float sm = 0.f;
        for (int b = 0; b < B; ++b) {
            float m = 0.f, T = 0.f;
            for (int k = 0; k < DM_CHUNKS; ++k) m += means_all[((size_t)s * B + b) * DM_CHUNKS + k];
            for (int k = 0; k < SI_CHUNKS; ++k) T += partial[((size_t)s * B + b) * SI_CHUNKS + k];
            const float inv = 1.f / (m / (float)hw + 1e-7f);
            aux[((size_t)s * B + b) * 2 + 0] = inv;
            aux[((size_t)s * B + b) * 2 + 1] = T;
            sm += (inv * T) * sample_w[b];
        }
        const float reg = smooth_scale / (float)(1 << s) * sm;
        losses[s * 4 + 1] = sm;
        losses[s * 4 + 2] = reg;
        losses[s * 4 + 3] += reg;
        add_total += reg;
    }
    losses[17] += add_total / 4.f;
}

__global__ __launch_bounds__(256) void smooth_intended_bwd_kernel(Pyramid pyr, SiPtrs im, const float* __restrict__ aux,
                                                                  const float* __restrict__ sample_w, SiDz out, int B,
                                                                  float smooth_scale) {
    const int b = blockIdx.y, sc = blockIdx.z;
    const int h = pyr.h[sc], w = pyr.w[sc], hw = h * w;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const float* d = pyr.disp[sc] + (size_t)b * hw;
    const float* img = im.rgb0[sc] + (size_t)b * 3 * hw;
    const float inv = aux[((size_t)sc * B + b) * 2 + 0], T = aux[((size_t)sc * B + b) * 2 + 1];
    const float nx = 1.f / (float)(h * (w - 1)), ny = 1.f / (float)((h - 1) * w);
    const int y = p / w, x = p - y * w;
    const float dp = d[p];
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    float g = 0.f;
    if (x + 1 < w) g += sgn(dp - d[p + 1]) * si_edge_weight(img, hw, p, p + 1) * nx;
    if (x >= 1) g -= sgn(d[p - 1] - dp) * si_edge_weight(img, hw, p - 1, p) * nx;
    if (y + 1 < h) g += sgn(dp - d[p + w]) * si_edge_weight(img, hw, p, p + w) * ny;
    if (y >= 1) g -= sgn(d[p - w] - dp) * si_edge_weight(img, hw, p - w, p) * ny;
    const float coef = smooth_scale / (float)(1 << sc) / 4.f * sample_w[b];
    const float dl = coef * (inv * g - T * inv * inv / (float)hw);
    out.dz[sc][(size_t)b * hw + p] += dp * (1.f - dp) * dl;      // through the sigmoid of the disparity head
}

}  // namespace clslam

extern "C" int clslam_smooth_intended_chunks() { return clslam::SI_CHUNKS; }

extern "C" int clslam_smooth_intended_fwd(const float* const* disp, const float* const* rgb0, float* partial, int batch, int H, int W,
                                          void* stream) {
    CLSLAM_REQUIRE(disp && rgb0 && partial && (H >> 3) >= 2 && (W >> 3) >= 2, "smooth_intended_fwd: bad args");
    if (!batch) return CLSLAM_OK;
    clslam::SiPtrs im;
    for (int s = 0; s < 4; ++s) im.rgb0[s] = rgb0[s];
    hipLaunchKernelGGL(clslam::smooth_intended_fwd_kernel, dim3(clslam::SI_CHUNKS, batch, 4), dim3(256), 0, (hipStream_t)stream,
                       make_pyramid(disp, H, W), im, partial, batch);
    return check_launch("smooth_intended_fwd");
}

extern "C" int clslam_smooth_intended_finalize(const float* partial, const float* means, const float* sample_w, float* losses,
                                               float* aux, int batch, int H, int W, float smooth_scale, void* stream) {
    CLSLAM_REQUIRE(partial && means && sample_w && losses && aux, "smooth_intended_finalize: null");
    if (!batch) return CLSLAM_OK;
    hipLaunchKernelGGL(clslam::smooth_intended_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, means, sample_w,
                       losses, aux, batch, H, W, smooth_scale);
    return check_launch("smooth_intended_finalize");
}

extern "C" int clslam_smooth_intended_bwd(const float* const* disp, const float* const* rgb0, const float* aux, const float* sample_w,
                                          float* const* dz, int batch, int H, int W, float smooth_scale, void* stream) {
    CLSLAM_REQUIRE(disp && rgb0 && aux && sample_w && dz, "smooth_intended_bwd: null");
    if (!batch) return CLSLAM_OK;
    clslam::SiPtrs im;
    clslam::SiDz out;
    for (int s = 0; s < 4; ++s) { im.rgb0[s] = rgb0[s]; out.dz[s] = dz[s]; }
    hipLaunchKernelGGL(clslam::smooth_intended_bwd_kernel, dim3(clslam::cdiv(H * W, 256), batch, 4), dim3(256), 0, (hipStream_t)stream,
                       make_pyramid(disp, H, W), im, aux, sample_w, out, batch, smooth_scale);
    return check_launch("smooth_intended_bwd");
}
