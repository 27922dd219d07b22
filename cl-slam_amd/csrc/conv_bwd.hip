// K15/K17: backward of the trainable convolutions (depth decoder, pose decoder).
// The reference gets these from autograd of networks/depth_decoder.py:51-71,
// networks/layers.py:9-48 and networks/pose_decoder.py:37-54 (cuDNN convolution_backward,
// reflection_pad2d_backward, upsample_nearest2d_backward, elu_backward, cat backward).
//
//   dgrad : clslam_weight_transpose + clslam_conv2d on the zero-padded (H+2)x(W+2) domain
//           (pad = 2) gives d(padded input); clslam_fold_act_grad folds the reflection border
//           back, sums the 2x2 nearest-upsample footprint and multiplies by act'(y).
//   wgrad : dW[n][tap][c] = sum_m dZ[m][n] * G(m,tap,c) as an MFMA GEMM whose reduction axis is
//           the pixel index m; split over pixel ranges (deterministic two-stage reduction:
//           clslam_wgrad writes per-split partials, clslam_reduce_partials sums them in order).
//   bias  : clslam_colsum (two-stage, deterministic).
#include "adam_dev.h"

namespace clslam {

// ------------------------------------------------------------------------------------------------
__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                        int taps, int Cin, int CinSel) {
    // wt[ci][taps-1-t][co] = w[co][t][ci]
    const int total = CinSel * taps * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout;
        const int t2 = (i / Cout) % taps;
        const int ci = i / (Cout * taps);
        wt[i] = w[((size_t)co * taps + (taps - 1 - t2)) * Cin + ci];
    }
}

// Up to kTransposeBatch weight tensors in one launch (table by value in the kernel arguments; blockIdx.y = item)
constexpr int kTransposeBatch = 16;
struct TransposeBatch {
    const float* w[kTransposeBatch];
    float* wt[kTransposeBatch];
    int cout[kTransposeBatch], taps[kTransposeBatch], cin[kTransposeBatch], cin_sel[kTransposeBatch];
};
__global__ void weight_transpose_multi_kernel(TransposeBatch b) {
    const int it = blockIdx.y;
    const float* __restrict__ w = b.w[it];
    float* __restrict__ wt = b.wt[it];
    const int Cout = b.cout[it], taps = b.taps[it], Cin = b.cin[it];
    const int total = b.cin_sel[it] * taps * Cout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Cout;
        const int t2 = (i / Cout) % taps;
        const int ci = i / (Cout * taps);
        wt[i] = w[((size_t)co * taps + (taps - 1 - t2)) * Cin + ci];
    }
}

// ------------------------------------------------------------------------------------------------
// dz[b,y,x,c] = act'(yout[b,y,x,c]) * sum_{(Y,X) in footprint(y,x)} sum_{P -> (Y,X)} dxp[b,P,c]
// disp_dz / disp_w (optional, pool == 0, e == 1): the dispconv head hanging off the same activation
// (depth_decoder.py:67-69).  Its data gradient on the padded domain, sum_k disp_dz[P-2+k] * disp_w[flip k][c],
// is evaluated here per folded position instead of a separate read-modify-write pass over dxp
// (clslam_dispconv_bwd_data); dxp may then be NULL (scale 0: no upconv feeds on this activation's padded grad).
__global__ __launch_bounds__(256) void fold_act_grad_kernel(const float* __restrict__ dxp, const float* __restrict__ yout,
                                                            float* __restrict__ dz, float* __restrict__ bias_partial, int B,
                                                            int H, int W, int C, int e, int pool, int act, int Cp,
                                                            const float* __restrict__ disp_dz, const float* __restrict__ disp_w) {
    // bias_partial (optional): [gridDim.x][C] per-block column sums of dz = the conv's bias gradient
    // partials (C/4 divides 256 and the grid stride, so a thread keeps one channel quad throughout).
    __shared__ float4 bred[256];
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
    // H, W: resolution of the conv input whose padded-domain gradient dxp [(H+2e)][(W+2e)][Cp] holds;
    // output resolution is (H,W) or (H/2,W/2) when pool; only channels [0,C) are consumed.
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    const int C4 = C / 4;
    const int Hp = H + 2 * e, Wp = W + 2 * e;
    // 32-bit index arithmetic (the host checks total < 2^31): 64-bit div/mod chains cost more than the loads
    const int total = B * Ho * Wo * C4;
    // C/4 divides 256 and the grid stride: a thread keeps one channel quad, so its 9 head weights stay in registers
    float4 dw[9];
    if (disp_dz) {
#pragma unroll
        for (int t = 0; t < 9; ++t) dw[t] = *reinterpret_cast<const float4*>(disp_w + t * C + (threadIdx.x % C4) * 4);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c4 = i % C4;
        const int pix = i / C4;
        const int x = pix % Wo;
        const int row = pix / Wo;
        const int y = row % Ho;
        const int b = row / Ho;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ny = pool ? 2 : 1;
        const int Yf = pool ? 2 * y : y, Xf = pool ? 2 * x : x;
        // the activation's output is requested together with the gradient taps below (it was a second, dependent round trip)
        const size_t o = ((size_t)(b * Ho + y) * Wo + x) * C + c4 * 4;
        float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (yout) yv = *reinterpret_cast<const float4*>(yout + o);
        // Fast path (all but the two outermost rows / columns): every footprint pixel folds from exactly one
        // padded position, so the 1 (or 2x2) float4 loads and the 9 head-gradient taps are unconditional and in
        // flight together.  The general path below walks variable-length lists: one load, one wait at a time.
        if (Yf >= 2 && Yf + ny - 1 <= H - 3 && Xf >= 2 && Xf + ny - 1 <= W - 3) {
            float4 v[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int dy = d >> 1, dx = d & 1;
                if (d < ny * ny && dxp)
                    v[d] = *reinterpret_cast<const float4*>(dxp + ((size_t)(b * Hp + Yf + dy + e) * Wp + Xf + dx + e) * Cp + c4 * 4);
                else
                    v[d] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (disp_dz) {      // pool == 0 here: the single padded position (Yf+1, Xf+1) sees the 3x3 neighbourhood of dz
                float g[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) g[t] = disp_dz[((size_t)b * H + Yf - 1 + t / 3) * W + Xf - 1 + t % 3];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float4 k = dw[8 - t];      // (2-ky)*3 + (2-kx)
                    v[0].x = fmaf(g[t], k.x, v[0].x); v[0].y = fmaf(g[t], k.y, v[0].y);
                    v[0].z = fmaf(g[t], k.z, v[0].z); v[0].w = fmaf(g[t], k.w, v[0].w);
                }
            }
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < ny * ny) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
        } else
        for (int dy = 0; dy < ny; ++dy) {
            for (int dx = 0; dx < ny; ++dx) {
                const int Y = pool ? 2 * y + dy : y, X = pool ? 2 * x + dx : x;
                int py[3], px[3];
                int npy = 0, npx = 0;
                py[npy++] = Y + e;
                px[npx++] = X + e;
                if (e) {
                    if (Y == 1) py[npy++] = 0;
                    if (Y == H - 2) py[npy++] = H + 1;
                    if (X == 1) px[npx++] = 0;
                    if (X == W - 2) px[npx++] = W + 1;
                }
                for (int a = 0; a < npy; ++a)
                    for (int q = 0; q < npx; ++q) {
                        if (dxp) {
                            const float4 v = *reinterpret_cast<const float4*>(
                                dxp + ((size_t)(b * Hp + py[a]) * Wp + px[q]) * Cp + c4 * 4);
                            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                        }
                        if (disp_dz) {
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky) {
                                const int yy = py[a] - 2 + ky;
                                if (yy < 0 || yy >= H) continue;
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx) {
                                    const int xx = px[q] - 2 + kx;
                                    if (xx < 0 || xx >= W) continue;
                                    const float g = disp_dz[((size_t)b * H + yy) * W + xx];
                                    const float4 k = dw[(2 - ky) * 3 + (2 - kx)];
                                    acc.x = fmaf(g, k.x, acc.x); acc.y = fmaf(g, k.y, acc.y);
                                    acc.z = fmaf(g, k.z, acc.z); acc.w = fmaf(g, k.w, acc.w);
                                }
                            }
                        }
                    }
            }
        }
        if (yout) {
            acc.x *= act_grad_from_output(yv.x, act);
            acc.y *= act_grad_from_output(yv.y, act);
            acc.z *= act_grad_from_output(yv.z, act);
            acc.w *= act_grad_from_output(yv.w, act);
        }
        *reinterpret_cast<float4*>(dz + o) = acc;
        bacc.x += acc.x; bacc.y += acc.y; bacc.z += acc.z; bacc.w += acc.w;
    }
    if (bias_partial) {
        // threads of one channel quad sit C4 lanes apart (C4 | 64): shuffle inside the wave, the four waves meet in LDS.  (The
        // serial walk over 256 / C4 LDS entries this replaces was 64 dependent reads per block at C = 16 -- on the
        // data-gradient chain, ten times per step.)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float4 v = bacc;
        for (int m = C4; m < 64; m <<= 1) {
            v.x += wave_shfl_xor(v.x, m); v.y += wave_shfl_xor(v.y, m);
            v.z += wave_shfl_xor(v.z, m); v.w += wave_shfl_xor(v.w, m);
        }
        if (lane < C4) bred[wave * 64 + lane] = v;
        __syncthreads();
        if ((int)threadIdx.x < C4) {
            const float4 a0 = bred[threadIdx.x], a1 = bred[64 + threadIdx.x], a2 = bred[128 + threadIdx.x], a3 = bred[192 + threadIdx.x];
            float4 t;
            t.x = (a0.x + a1.x) + (a2.x + a3.x); t.y = (a0.y + a1.y) + (a2.y + a3.y);
            t.z = (a0.z + a1.z) + (a2.z + a3.z); t.w = (a0.w + a1.w) + (a2.w + a3.w);
            *reinterpret_cast<float4*>(bias_partial + (size_t)blockIdx.x * C + threadIdx.x * 4) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct WgradK {
    const float* __restrict__ dz;
    const float* __restrict__ src_a;
    const float* __restrict__ src_b;
    float* __restrict__ partial;
    int B, Hi, Wi, Ca, Cb, Ho, Wo, Cout;
    int ksize, stride, pad, pad_mode, ups;
    int M, chunks_per_split, co_tiles, ci_tiles, tap_groups;
};

template <int COT, int CIT, int NT, int MF>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradK p) {
    constexpr int KM = 32;                       // pixels per chunk
    constexpr int LDZ = COT + 4, LDG = CIT + 4;  // float4-aligned rows
    constexpr int TI = COT / MF, TJ = CIT / MF;
    constexpr int NTILES = TI * TJ * NT;
    constexpr int TPW = (NTILES + 3) / 4;        // MFMA tiles per wave
    constexpr int NACC = MF == 32 ? 16 : 4;
    constexpr int KG = 64 / MF;                  // pixel rows consumed per MFMA
    constexpr int Z_F4 = KM * COT / 4, G_F4 = KM * CIT / 4;
    constexpr int Z_IT = (Z_F4 + 255) / 256, G_IT = (G_F4 + 255) / 256;

    __shared__ __attribute__((aligned(16))) float Zs[KM * LDZ];
    __shared__ __attribute__((aligned(16))) float Gs[NT][KM * LDG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx = blockIdx.x;
    const int cot = bx % p.co_tiles; bx /= p.co_tiles;
    const int cit = bx % p.ci_tiles; bx /= p.ci_tiles;
    const int tg = bx;
    const int split = blockIdx.y;
    const int co0 = cot * COT, ci0 = cit * CIT, t0 = tg * NT;
    const int Cin = p.Ca + p.Cb, taps = p.ksize * p.ksize;
    const int HA = p.ups ? (p.Hi >> 1) : p.Hi, WA = p.ups ? (p.Wi >> 1) : p.Wi;

    typedef float accv __attribute__((ext_vector_type(NACC)));
    accv acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[s][r] = 0.f;

    const int mbeg = split * p.chunks_per_split * KM;
    const int mend = min(p.M, mbeg + p.chunks_per_split * KM);

    // Software pipeline: chunk mc+KM is fetched global->registers while the MFMAs of chunk mc run; the
    // zero masks are applied when the registers are written to LDS (see conv_patch.hip).
    float4 rz[Z_IT], rg[NT][G_IT];
    bool okz_[Z_IT], okg_[NT][G_IT];
    auto load_chunk = [&](int mc) {
#pragma unroll
        for (int it = 0; it < Z_IT; ++it) {
            const int f = tid + it * 256;
            const int row = f / (COT / 4), c4 = f % (COT / 4);
            const int m = mc + row;
            const bool okz = (f < Z_F4) && (m < mend);
            rz[it] = *reinterpret_cast<const float4*>(p.dz + (size_t)(okz ? m : 0) * p.Cout + co0 + (okz ? c4 * 4 : 0));
            okz_[it] = okz;
        }
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int f = tid + it * 256;
            const int row = f / (CIT / 4), c4 = f % (CIT / 4);
            const int m = mc + row;
            const bool okm = (f < G_F4) && (m < mend);
            const int mm = okm ? m : 0;
            const int b = mm / (p.Ho * p.Wo);
            const int r = mm - b * (p.Ho * p.Wo);
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            const int c = ci0 + c4 * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tap = t0 + t;
                const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
                int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                bool ok = okm && tap < taps;
                if (p.pad_mode == CLSLAM_PAD_REFLECT) {
                    iy = reflect_idx(iy, p.Hi);
                    ix = reflect_idx(ix, p.Wi);
                } else {
                    ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
                }
                const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                const size_t oa = ok ? ((size_t)(b * HA + sy) * WA + sx) * p.Ca : 0;
                const size_t ob = ok ? ((size_t)(b * p.Hi + iy) * p.Wi + ix) * p.Cb : 0;
                const float* ptr = (c < p.Ca) ? p.src_a + oa + c : p.src_b + ob + (c - p.Ca);
                rg[t][it] = *reinterpret_cast<const float4*>(ptr);   // unconditional, masked at store time
                okg_[t][it] = ok;
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < Z_IT; ++it) {
            const int f = tid + it * 256;
            float4 v = rz[it];
            if (!okz_[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < Z_F4) *reinterpret_cast<float4*>(&Zs[(f / (COT / 4)) * LDZ + (f % (COT / 4)) * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < G_IT; ++it) {
            const int f = tid + it * 256;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float4 v = rg[t][it];
                if (!okg_[t][it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (f < G_F4) *reinterpret_cast<float4*>(&Gs[t][(f / (CIT / 4)) * LDG + (f % (CIT / 4)) * 4]) = v;
            }
        }
    };

    if (mbeg < mend) load_chunk(mbeg);
    for (int mc = mbeg; mc < mend; mc += KM) {
        __syncthreads();  // previous chunk's MFMAs are done reading LDS
        store_chunk();
        __syncthreads();
        if (mc + KM < mend) load_chunk(mc + KM);
        // ---- MFMA over the KM pixels of the chunk -------------------------------------------
#pragma unroll
        for (int ks = 0; ks < KM / KG; ++ks) {
            const int row = ks * KG + lane / MF;
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                const int q = wave + 4 * s;
                if (q < NTILES) {
                    const int ti = q % TI, tj = (q / TI) % TJ, t = q / (TI * TJ);
                    const float a = Zs[row * LDZ + ti * MF + lane % MF];
                    const float b = Gs[t][row * LDG + tj * MF + lane % MF];
                    if constexpr (MF == 32) acc[s] = mfma_32x32x2(a, b, acc[s]);
                    else acc[s] = mfma_16x16x4(a, b, acc[s]);
                }
            }
        }
    }

    // ---- write this split's partial tile: partial[split][co][tap][ci] -------------------------
    float* out = p.partial + (size_t)split * p.Cout * taps * Cin;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int q = wave + 4 * s;
        if (q >= NTILES) continue;
        const int ti = q % TI, tj = (q / TI) % TJ, t = q / (TI * TJ);
        const int tap = t0 + t;
        if (tap >= taps) continue;
        const int ci = ci0 + tj * MF + lane % MF;
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            int row;
            if constexpr (MF == 32) row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            else row = 4 * (lane >> 4) + r;
            const int co = co0 + ti * MF + row;
            out[((size_t)co * taps + tap) * Cin + ci] = acc[s][r];
        }
    }
}

template <int COT, int CIT, int NT, int MF>
static int launch_wgrad(WgradK k, int splits, hipStream_t stream) {
    const int taps = k.ksize * k.ksize;
    k.co_tiles = k.Cout / COT;
    k.ci_tiles = (k.Ca + k.Cb) / CIT;
    k.tap_groups = cdiv(taps, NT);
    hipLaunchKernelGGL((conv_wgrad_kernel<COT, CIT, NT, MF>), dim3(k.co_tiles * k.ci_tiles * k.tap_groups, splits),
                       dim3(256), 0, stream, k);
    return check_launch("conv_wgrad");
}

// out[i] = scale * sum_s partial[s][i].  IL outputs x KL split-lanes per block: lane kl sums splits
// kl, kl+KL, ... in order, then the KL partial sums are added in order (deterministic; also fast when
// n is tiny and splits is large, e.g. bias gradients).
template <int IL>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             size_t n, int splits, float scale) {
    constexpr int KL = 256 / IL;
    __shared__ float red[256];
    const int il = threadIdx.x % IL, kl = threadIdx.x / IL;
    for (size_t i0 = (size_t)blockIdx.x * IL; i0 < n; i0 += (size_t)gridDim.x * IL) {
        const size_t i = i0 + il;
        float s = 0.f;
        if (i < n)
            for (int k = kl; k < splits; k += KL) s += partial[(size_t)k * n + i];
        red[threadIdx.x] = s;
        __syncthreads();
        if (kl == 0 && i < n) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < KL; ++q) t += red[q * IL + il];
            out[i] = t * scale;
        }
        __syncthreads();
    }
}

// Batched form: one launch reduces a whole table of (partial, out, n, splits) items (all the weight /
// bias gradients of a step); blockIdx.y = item, blockIdx.x strides over the item's outputs.
struct ReduceItem { const float* partial; float* out; unsigned long long n; int splits; float scale; };

// ADAM: the optimizer step (adam.hip's operation order, torch.optim.Adam's) is applied to every reduced element on the spot --
// single-GPU path only, where nothing sits between the gradient reduction and the update: one launch and one pass over
// the gradient less per step.  The gradient arena is still written (callers and tests read it).
struct AdamArgs {
    const float* g_base;     // gradient arena; an item's `out` lies inside it, the same offset addresses p / m / v
    float* p; float* m; float* v;
    float step_size, w1, beta2, w2, bc2_sqrt, eps;
    const float* guard;      // the step's loss: NaN -> gradients are reduced, nothing is updated
};

__device__ __forceinline__ void adam_apply(const AdamArgs& a, size_t idx, float gr) {
    float pk = a.p[idx], mk = a.m[idx], vk = a.v[idx];
    adam_update(pk, mk, vk, gr, a.step_size, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps);
    a.p[idx] = pk; a.m[idx] = mk; a.v[idx] = vk;
}

__device__ __forceinline__ void adam_apply4(const AdamArgs& a, size_t idx, const float4& g4) {   // idx: multiple of 4
    float4 pp = *reinterpret_cast<float4*>(a.p + idx), mm = *reinterpret_cast<float4*>(a.m + idx), vv = *reinterpret_cast<float4*>(a.v + idx);
    float* P = &pp.x; float* M = &mm.x; float* V = &vv.x;
    const float G[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_update(P[k], M[k], V[k], G[k], a.step_size, a.w1, a.beta2, a.w2, a.bc2_sqrt, a.eps);
    *reinterpret_cast<float4*>(a.p + idx) = pp; *reinterpret_cast<float4*>(a.m + idx) = mm; *reinterpret_cast<float4*>(a.v + idx) = vv;
}

template <bool ADAM>
__global__ __launch_bounds__(256) void reduce_multi_kernel(const ReduceItem* __restrict__ items, AdamArgs ad) {
    __shared__ double redd[256][4];
    const bool update = ADAM && !(ad.guard && !(ad.guard[0] == ad.guard[0]));
    const ReduceItem it = items[blockIdx.y];
    // Lanes along the split axis (KL) by split count: the big-weight layers (most of the bytes) have <= 24
    // splits -> one thread per float4 output sums all of them from independent loads in flight, no LDS, no
    // barrier; layers with hundreds of splits and few outputs (16/32-channel convs, bias gradients) spend
    // 4 or 16 lanes per output on the split axis and combine them through LDS in lane order.
    const int KL = it.splits <= 24 ? ((it.n & 3) ? 4 : 1) : (it.splits <= 96 ? 4 : (it.splits <= 384 ? 16 : 64));
    const int IL = 256 / KL;
    const int il = threadIdx.x % IL, kl = threadIdx.x / IL;
    if ((it.n & 3) == 0) {   // 16-byte path (all conv weights / biases)
        const size_t n4 = it.n >> 2;
        if (KL == 1) {
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
                const float4* src = reinterpret_cast<const float4*>(it.partial) + i;
                // the split partials are summed in DOUBLE and rounded once (round 5): with up to several hundred fp32 partials per
                // element the fp32 running sum was the largest single source of elementwise gradient noise -- what decides the
                // SIGN of the near-zero entries, i.e. Adam's first lr * sign(g) updates (tests/test_teacher_forced_steps.py)
                double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
                int k = 0;
                for (; k + 4 <= it.splits; k += 4) {   // four independent loads in flight, summed in split order
                    const float4 v0 = src[(size_t)k * n4], v1 = src[(size_t)(k + 1) * n4];
                    const float4 v2 = src[(size_t)(k + 2) * n4], v3 = src[(size_t)(k + 3) * n4];
                    sx += v0.x; sy += v0.y; sz += v0.z; sw += v0.w;
                    sx += v1.x; sy += v1.y; sz += v1.z; sw += v1.w;
                    sx += v2.x; sy += v2.y; sz += v2.z; sw += v2.w;
                    sx += v3.x; sy += v3.y; sz += v3.z; sw += v3.w;
                }
                for (; k < it.splits; ++k) {
                    const float4 v = src[(size_t)k * n4];
                    sx += v.x; sy += v.y; sz += v.z; sw += v.w;
                }
                float4 s = make_float4((float)sx, (float)sy, (float)sz, (float)sw);
                s.x *= it.scale; s.y *= it.scale; s.z *= it.scale; s.w *= it.scale;
                reinterpret_cast<float4*>(it.out)[i] = s;
                if (ADAM && update) adam_apply4(ad, (size_t)(it.out - ad.g_base) + i * 4, s);
            }
            return;
        }
        for (size_t i0 = (size_t)blockIdx.x * IL; i0 < n4; i0 += (size_t)gridDim.x * IL) {
            const size_t i = i0 + il;
            double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
            if (i < n4)
                for (int k = kl; k < it.splits; k += KL) {
                    const float4 v = reinterpret_cast<const float4*>(it.partial + (size_t)k * it.n)[i];
                    sx += v.x; sy += v.y; sz += v.z; sw += v.w;
                }
            redd[threadIdx.x][0] = sx; redd[threadIdx.x][1] = sy; redd[threadIdx.x][2] = sz; redd[threadIdx.x][3] = sw;
            __syncthreads();
            if (kl == 0 && i < n4) {
                double tx = redd[il][0], ty = redd[il][1], tz = redd[il][2], tw = redd[il][3];
                for (int q = 1; q < KL; ++q) { const double* v = redd[q * IL + il]; tx += v[0]; ty += v[1]; tz += v[2]; tw += v[3]; }
                float4 t = make_float4((float)tx, (float)ty, (float)tz, (float)tw);
                t.x *= it.scale; t.y *= it.scale; t.z *= it.scale; t.w *= it.scale;
                reinterpret_cast<float4*>(it.out)[i] = t;
                if (ADAM && update) adam_apply4(ad, (size_t)(it.out - ad.g_base) + i * 4, t);
            }
            __syncthreads();
        }
        return;
    }
    for (size_t i0 = (size_t)blockIdx.x * IL; i0 < it.n; i0 += (size_t)gridDim.x * IL) {
        const size_t i = i0 + il;
        double s = 0.0;
        if (i < it.n)
            for (int k = kl; k < it.splits; k += KL) s += it.partial[(size_t)k * it.n + i];
        redd[threadIdx.x][0] = s;
        __syncthreads();
        if (kl == 0 && i < it.n) {
            double td = redd[il][0];
            for (int q = 1; q < KL; ++q) td += redd[q * IL + il][0];
            const float t = (float)td;
            it.out[i] = t * it.scale;
            if (ADAM && update) adam_apply(ad, (size_t)(it.out - ad.g_base) + i, t * it.scale);
        }
        __syncthreads();
    }
}

// partial[blk][c] = sum over the block's rows of x[row][c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                     int M, int C, int rows_per_block) {
    __shared__ float red[256];
    const int lanes = 256 / C;  // row lanes (C <= 256)
    const int c = threadIdx.x % C, rl = threadIdx.x / C;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (rl < lanes)
        for (int r = r0 + rl; r < r1; r += lanes) s += x[(size_t)r * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < lanes; ++k) t += red[k * C + threadIdx.x];
        partial[(size_t)blockIdx.x * C + threadIdx.x] = t;
    }
}

}  // namespace clslam

using namespace clslam;

extern "C" int clslam_weight_transpose(const float* w, float* wt, int ch_out, int taps, int ch_in, int ch_in_sel,
                                       void* stream) {
    CLSLAM_REQUIRE(w && wt && ch_in_sel <= ch_in, "weight_transpose: bad args");
    const int total = ch_in_sel * taps * ch_out;
    if (total == 0) return CLSLAM_OK;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(min(1024, cdiv(total, 256))), dim3(256), 0, (hipStream_t)stream,
                       w, wt, ch_out, taps, ch_in, ch_in_sel);
    return check_launch("weight_transpose");
}

extern "C" int clslam_weight_transpose_multi(const clslam_transpose_item* items, int nitems, void* stream) {
    CLSLAM_REQUIRE(nitems >= 0 && (items || nitems == 0), "weight_transpose_multi: bad args");
    for (int base = 0; base < nitems; base += kTransposeBatch) {
        TransposeBatch b;
        int n = 0, largest = 0;
        for (int i = base; i < nitems && n < kTransposeBatch; ++i) {
            const clslam_transpose_item& t = items[i];
            CLSLAM_REQUIRE(t.w && t.wt && t.ch_in_sel <= t.ch_in && t.ch_out > 0 && t.taps > 0, "weight_transpose_multi: bad item %d", i);
            if (t.ch_in_sel == 0) continue;
            b.w[n] = t.w; b.wt[n] = t.wt; b.cout[n] = t.ch_out; b.taps[n] = t.taps; b.cin[n] = t.ch_in; b.cin_sel[n] = t.ch_in_sel;
            largest = std::max(largest, t.ch_in_sel * t.taps * t.ch_out);
            ++n;
        }
        if (!n) continue;
        hipLaunchKernelGGL(weight_transpose_multi_kernel, dim3(std::min(256, cdiv(largest, 256)), n), dim3(256), 0,
                           (hipStream_t)stream, b);
        const int rc = check_launch("weight_transpose_multi");
        if (rc != CLSLAM_OK) return rc;
    }
    return CLSLAM_OK;
}

extern "C" int clslam_fold_blocks(int batch, int h, int w, int ch, int pool) {
    const size_t total = (size_t)batch * (pool ? h / 2 : h) * (pool ? w / 2 : w) * (ch / 4);
    // more workgroups do not help (measured 1024..8192: 33 -> 39 us at 192x640x16): the pass is HBM read+write bound
    return (int)std::max<size_t>(1, std::min<size_t>(1024, (total + 255) / 256));
}

extern "C" int clslam_fold_act_grad(const float* dxp, const float* yout, float* dz, float* bias_partial, int batch, int h,
                                    int w, int ch, int ch_stride, int border, int pool, int act, const float* disp_dz,
                                    const float* disp_w, void* stream) {
    CLSLAM_REQUIRE((dxp || disp_dz) && dz && ch % 4 == 0 && ch_stride % 4 == 0 && ch <= ch_stride, "fold_act_grad: bad args");
    CLSLAM_REQUIRE(!disp_dz || (disp_w && !pool && border == 1 && 256 % (ch / 4) == 0),
                   "fold_act_grad: the fused dispconv gradient needs disp_w, border 1, no pooling and ch/4 dividing 256");
    CLSLAM_REQUIRE(border == 0 || border == 1, "fold_act_grad: border must be 0/1");
    CLSLAM_REQUIRE(!pool || (h % 2 == 0 && w % 2 == 0), "fold_act_grad: pooling needs even dims");
    const size_t total = (size_t)batch * (pool ? h / 2 : h) * (pool ? w / 2 : w) * (ch / 4);
    if (total == 0) return CLSLAM_OK;
    CLSLAM_REQUIRE(!bias_partial || (64 % (ch / 4) == 0), "fold_act_grad: fused bias sums need ch/4 to divide 64");
    CLSLAM_REQUIRE(total < ((size_t)1 << 31) - 256 * 4096, "fold_act_grad: tensor too large for 32-bit indexing");
    const int blocks = clslam_fold_blocks(batch, h, w, ch, pool);
#if CLSLAM_DEVICE_BUILD
    if (hipEvent_t done = take_handoff_event()) {      // the weight-gradient stream is released by this launch's own completion
        hipExtLaunchKernelGGL(fold_act_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nullptr, done, 0, dxp, yout, dz,
                              bias_partial, batch, h, w, ch, border, pool, act, ch_stride, disp_dz, disp_w);
        return check_launch("fold_act_grad");
    }
#endif
    hipLaunchKernelGGL(fold_act_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dxp, yout, dz, bias_partial, batch,
                       h, w, ch, border, pool, act, ch_stride, disp_dz, disp_w);
    return check_launch("fold_act_grad");
}

static void pick_wgrad_tile(const clslam_conv_desc* d, int* cot, int* nt) {
    const int Cin = d->ch_a + d->ch_b, taps = d->ksize * d->ksize;
    const bool c64 = d->ch_out % 64 == 0 && Cin % 64 == 0 && (d->ch_b == 0 || d->ch_a % 64 == 0);
    const bool c32 = d->ch_out % 32 == 0 && Cin % 32 == 0 && (d->ch_b == 0 || d->ch_a % 32 == 0);
    if (c64) { *cot = 64; *nt = taps == 9 ? 3 : 1; }
    else if (c32) { *cot = 32; *nt = taps == 9 ? 9 : 1; }
    else { *cot = 16; *nt = taps == 9 ? 9 : 1; }
}

// Number of pixel-range splits clslam_conv_wgrad should be launched with so that the grid has
// about target_blocks workgroups (the caller sizes the partial buffer from it).
extern "C" int clslam_wgrad_splits(const clslam_conv_desc* d, int target_blocks) {
    const int M = d->batch * d->out_h * d->out_w;
    const int Cin = d->ch_a + d->ch_b, taps = d->ksize * d->ksize;
    int T, NT;
    pick_wgrad_tile(d, &T, &NT);
    const int tiles = (d->ch_out / T) * (Cin / T) * cdiv(taps, NT);
    const int chunks = std::max(1, cdiv(M, 32));
    const int splits = std::max(1, std::min(chunks, cdiv(target_blocks, tiles)));
    const int cps = cdiv(chunks, splits);
    return cdiv(chunks, cps);
}

// dW partials.  desc describes the FORWARD conv (src_a/src_b/geometry); dz = d(pre-activation
// output) [B*out_h*out_w][ch_out]; partial holds splits*ch_out*taps*(ch_a+ch_b) floats.
extern "C" int clslam_conv_wgrad(const clslam_conv_desc* d, const float* dz, float* partial, int splits, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CLSLAM_REQUIRE(d && dz && partial && d->src_a && splits >= 1, "conv_wgrad: bad args");
    const int Cin = d->ch_a + d->ch_b, taps = d->ksize * d->ksize;
    CLSLAM_REQUIRE(taps == 1 || taps == 9, "conv_wgrad: ksize must be 1 or 3");
    CLSLAM_REQUIRE(d->ch_out % 16 == 0 && Cin % 16 == 0 && d->ch_a % 16 == 0, "conv_wgrad: channels must be multiples of 16");
    WgradK k;
    k.dz = dz; k.src_a = d->src_a; k.src_b = d->src_b; k.partial = partial;
    k.B = d->batch; k.Hi = d->in_h; k.Wi = d->in_w; k.Ca = d->ch_a; k.Cb = d->ch_b; k.Ho = d->out_h; k.Wo = d->out_w;
    k.Cout = d->ch_out; k.ksize = d->ksize; k.stride = d->stride; k.pad = d->pad; k.pad_mode = d->pad_mode;
    k.ups = d->upsample_a;
    k.M = d->batch * d->out_h * d->out_w;
    const int chunks = std::max(1, cdiv(k.M, 32));
    k.chunks_per_split = cdiv(chunks, splits);
    k.co_tiles = k.ci_tiles = k.tap_groups = 0;
    int T, NT;
    pick_wgrad_tile(d, &T, &NT);
    if (T == 64) return NT == 3 ? launch_wgrad<64, 64, 3, 32>(k, splits, stream) : launch_wgrad<64, 64, 1, 32>(k, splits, stream);
    if (T == 32) return NT == 9 ? launch_wgrad<32, 32, 9, 32>(k, splits, stream) : launch_wgrad<32, 32, 1, 32>(k, splits, stream);
    return NT == 9 ? launch_wgrad<16, 16, 9, 16>(k, splits, stream) : launch_wgrad<16, 16, 1, 16>(k, splits, stream);
}

extern "C" int clslam_reduce_partials(const float* partial, float* out, size_t n, int splits, float scale, void* stream) {
    CLSLAM_REQUIRE(partial && out, "reduce_partials: null");
    if (n == 0) return CLSLAM_OK;
    if (n >= 16384 || splits <= 8) {
        const int blocks = (int)std::min<size_t>(4096, (n + 63) / 64);
        hipLaunchKernelGGL(reduce_partials_kernel<64>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, out, n, splits, scale);
    } else {
        const int blocks = (int)std::min<size_t>(4096, (n + 15) / 16);
        hipLaunchKernelGGL(reduce_partials_kernel<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, out, n, splits, scale);
    }
    return check_launch("reduce_partials");
}

// items_dev: device array of nitems {const float* partial; float* out; uint64 n; int32 splits; float scale}
// (32 bytes each).  One launch for all of a step's gradient reductions.
extern "C" int clslam_reduce_multi(const void* items_dev, int nitems, int blocks_per_item, void* stream) {
    CLSLAM_REQUIRE(items_dev && nitems >= 0 && blocks_per_item >= 1, "reduce_multi: bad args");
    static_assert(sizeof(ReduceItem) == 32, "ReduceItem layout");
    if (!nitems) return CLSLAM_OK;
    hipLaunchKernelGGL(reduce_multi_kernel<false>, dim3(blocks_per_item, nitems), dim3(256), 0, (hipStream_t)stream,
                       (const ReduceItem*)items_dev, AdamArgs{});
    return check_launch("reduce_multi");
}

// reduce_multi + the optimizer step on every reduced element (clslam_adam_step's arithmetic; scalars formed in double).
// Every trainable element must be the output of exactly one item (gradients written elsewhere: an item with
// partial == out, splits = 1).  The 16-byte path needs the arenas' bases and every item's offset 16-byte aligned.
extern "C" int clslam_reduce_multi_adam(const void* items_dev, int nitems, int blocks_per_item, const float* grad_base,
                                        float* param, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                                        double eps, int step, const float* guard, void* stream) {
    CLSLAM_REQUIRE(items_dev && nitems >= 0 && blocks_per_item >= 1 && grad_base && param && exp_avg && exp_avg_sq && step >= 1,
                   "reduce_multi_adam: bad args");
    CLSLAM_REQUIRE(((((uintptr_t)grad_base) | ((uintptr_t)param) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15) == 0,
                   "reduce_multi_adam: arenas must be 16-byte aligned");
    if (!nitems) return CLSLAM_OK;
    AdamArgs ad;
    ad.g_base = grad_base; ad.p = param; ad.m = exp_avg; ad.v = exp_avg_sq;
    ad.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    ad.w1 = (float)(1.0 - beta1); ad.beta2 = (float)beta2; ad.w2 = (float)(1.0 - beta2);
    ad.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step)); ad.eps = (float)eps; ad.guard = guard;
    hipLaunchKernelGGL(reduce_multi_kernel<true>, dim3(blocks_per_item, nitems), dim3(256), 0, (hipStream_t)stream,
                       (const ReduceItem*)items_dev, ad);
    return check_launch("reduce_multi_adam");
}

extern "C" int clslam_colsum_blocks(int rows) { return std::max(1, std::min(512, cdiv(rows, 16))); }

// partial must hold clslam_colsum_blocks(rows)*ch floats; follow with clslam_reduce_partials.
extern "C" int clslam_colsum(const float* x, float* partial, int rows, int ch, void* stream) {
    CLSLAM_REQUIRE(x && partial && ch >= 1 && ch <= 256, "colsum: ch must be in [1,256]");
    const int blocks = clslam_colsum_blocks(rows);
    const int rpb = cdiv(std::max(rows, 1), blocks);
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, partial, rows, ch, rpb);
    return check_launch("colsum");
}
