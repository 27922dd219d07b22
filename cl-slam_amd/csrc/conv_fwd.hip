// K3/K4/K5: fp32 implicit-GEMM convolution on the CDNA4 matrix cores.
//
//   out[m][n] = act( scale[n] * sum_{tap,c} G(m,tap,c) * W[n][tap][c] + shift[n] (+ residual[m][n]) )
//
// m = (b, oy, ox) output pixel, n = output channel, NHWC activations, OHWI weights.  The gather
// G resolves, inside the A-operand loader, everything the reference does with separate ops:
//   * zero padding (torchvision BasicBlock convs, pose decoder; reference resnet_encoder.py:118-125,
//     pose_decoder.py:28-30), any stride,
//   * ReflectionPad2d(1) (reference layers.py:39-48),
//   * nearest 2x upsampling of source A and channel-concat with skip source B
//     (reference depth_decoder.py:56-65),
// and the epilogue fuses folded eval-mode BatchNorm (scale/shift), conv bias (shift), the residual
// add and ReLU / ELU.  The same kernel computes data gradients (dgrad = conv with transposed,
// flipped weights over a zero-padded domain; see conv_bwd.hip).
//
// Tiling: 256 threads = 4 wavefronts (2x2 or 4x1); block tile BM x BN, K chunk BK channels of one tap;
// A/B chunks are staged global -> registers -> LDS (rows padded to BK+4 floats so the
// ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct 4-bank slots), double
// buffered with one barrier per chunk; each wave owns TM x TN MFMA tiles of MF x MF
// (v_mfma_f32_32x32x2_f32 or v_mfma_f32_16x16x4_f32).  A lane's float4 LDS read supplies four
// consecutive k-steps (k = 4*(lane/MF)+t), so no operand shuffling is needed.
#include "common.h"

#include <type_traits>

namespace clslam {

struct ConvK {
    const float* __restrict__ src_a;
    const float* __restrict__ src_b;
    const float* __restrict__ wgt;
    const float* __restrict__ scale;
    const float* __restrict__ shift;
    const float* __restrict__ residual;
    const float* __restrict__ actgrad_src;
    float* __restrict__ out;
    int actgrad_kind;
    int B, Hi, Wi, Ca, Cb, Ho, Wo, Cout;
    int ksize, stride, pad, pad_mode, ups, act;
    int M, tilesM, tilesN, nblk;
};

template <int BM, int BN, int BK, int MF, int WGM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvK p) {
    constexpr int LDA = BK + 4;
    constexpr int WGN = 4 / WGM;                  // wave grid WGM x WGN (4 waves)
    constexpr int WM = BM / WGM, WN = BN / WGN;   // per-wave tile
    constexpr int TM = WM / MF, TN = WN / MF;     // MFMA tiles per wave
    constexpr int KG = 64 / MF;                   // k-groups inside a wave (2 for 32x32x2, 4 for 16x16x4)
    constexpr int KSTEP = KG * 4;                 // k consumed per float4 fragment read
    constexpr int F4_ROW = BK / 4;                // float4 per tile row
    constexpr int A_IT = (BM * F4_ROW + 255) / 256;
    constexpr int B_IT = (BN * F4_ROW + 255) / 256;
    constexpr int NACC = MF == 32 ? 16 : 4;
    static_assert(WM % MF == 0 && WN % MF == 0 && BK % KSTEP == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDA];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;

    const int logical = xcd_remap((int)blockIdx.x, p.nblk);
    const int tn = logical / p.tilesM, tm = logical - tn * p.tilesM;
    const int m0 = tm * BM, n0 = tn * BN;

    const int Cin = p.Ca + p.Cb;
    const int taps = p.ksize * p.ksize;
    const int HA = p.ups ? (p.Hi >> 1) : p.Hi, WA = p.ups ? (p.Wi >> 1) : p.Wi;

    // ---- per-thread A-load slots: fixed output pixel per slot --------------------------------
    int a_b[A_IT], a_iy0[A_IT], a_ix0[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int f = tid + it * 256;
        const int row = f / F4_ROW;
        const int m = m0 + row;
        a_ok[it] = (f < BM * F4_ROW) && (m < p.M);
        const int mm = a_ok[it] ? m : 0;
        const int b = mm / (p.Ho * p.Wo);
        const int r = mm - b * (p.Ho * p.Wo);
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        a_b[it] = b;
        a_iy0[it] = oy * p.stride - p.pad;
        a_ix0[it] = ox * p.stride - p.pad;
    }

    float4 ra[A_IT], rb[B_IT];
    bool ra_ok[A_IT];   // zero masks of the chunk in flight, applied when it is written to LDS
    const int chunks_per_tap = Cin / BK;
    const int niter = taps * chunks_per_tap;

    auto load_global = [&](int iter) {
        const int tap = iter / chunks_per_tap;
        const int c0 = (iter - tap * chunks_per_tap) * BK;
        const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int f = tid + it * 256;
            const int c = c0 + (f % F4_ROW) * 4;
            int iy = a_iy0[it] + ky, ix = a_ix0[it] + kx;
            bool ok = a_ok[it];
            if (p.pad_mode == CLSLAM_PAD_REFLECT) {
                iy = reflect_idx(iy, p.Hi);
                ix = reflect_idx(ix, p.Wi);
            } else {
                ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            }
            // unconditional load from a valid address; the zero mask is applied in store_lds, after the
            // MFMAs of the current chunk (a branch around the load, or a mask right behind it, makes the
            // compiler wait vmcnt(0) before the MFMA block and serialises the prefetch)
            const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
            const size_t oa = ok ? ((size_t)(a_b[it] * HA + sy) * WA + sx) * p.Ca : 0;
            const size_t ob = ok ? ((size_t)(a_b[it] * p.Hi + iy) * p.Wi + ix) * p.Cb : 0;
            const float* ptr = (c < p.Ca) ? p.src_a + oa + c : p.src_b + ob + (c - p.Ca);
            ra[it] = *reinterpret_cast<const float4*>(ptr);
            ra_ok[it] = ok;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int f = tid + it * 256;
            const int row = f / F4_ROW;
            const int n = n0 + row;
            const bool okb = (f < BN * F4_ROW) && (n < p.Cout);
            rb[it] = *reinterpret_cast<const float4*>(p.wgt + ((size_t)(okb ? n : 0) * taps + tap) * Cin + c0 + (f % F4_ROW) * 4);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int f = tid + it * 256;
            float4 v = ra[it];
            if (!ra_ok[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < BM * F4_ROW)
                *reinterpret_cast<float4*>(&As[buf][(f / F4_ROW) * LDA + (f % F4_ROW) * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int f = tid + it * 256;
            float4 v = rb[it];
            if (n0 + f / F4_ROW >= p.Cout) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < BN * F4_ROW)
                *reinterpret_cast<float4*>(&Bs[buf][(f / F4_ROW) * LDA + (f % F4_ROW) * 4]) = v;
        }
    };

    typedef float accv __attribute__((ext_vector_type(NACC)));
    accv acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) acc[i][j][r] = 0.f;

    const int frow = lane % MF;        // row inside an MFMA tile this lane feeds
    const int kg = lane / MF;          // k-group

    load_global(0);
    store_lds(0);
    __syncthreads();

    for (int iter = 0; iter < niter; ++iter) {
        const int buf = iter & 1;
        if (iter + 1 < niter) load_global(iter + 1);
#pragma unroll
        for (int kk = 0; kk < BK / KSTEP; ++kk) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const float4*>(&As[buf][(wm0 + i * MF + frow) * LDA + kk * KSTEP + kg * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const float4*>(&Bs[buf][(wn0 + j * MF + frow) * LDA + kk * KSTEP + kg * 4]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = t == 0 ? fa[i].x : t == 1 ? fa[i].y : t == 2 ? fa[i].z : fa[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = t == 0 ? fb[j].x : t == 1 ? fb[j].y : t == 2 ? fb[j].z : fb[j].w;
                        if constexpr (MF == 32) acc[i][j] = mfma_32x32x2(av, bv, acc[i][j]);
                        else acc[i][j] = mfma_16x16x4(av, bv, acc[i][j]);
                    }
                }
            }
        }
        if (iter + 1 < niter) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: BN/bias, residual, activation, NHWC store (lanes run along channels) ------
    // Straight-line phases (operands of all elements from clamped addresses, values with the activation switch outside the
    // element loop, then the stores back to back): see conv_patch.hip -- the element-by-element form serialised every store
    // behind an `s_waitcnt vmcnt(0)`.
    auto element = [&](int i, int j, int r, bool& ok) -> size_t {
        int row;
        if constexpr (MF == 32) row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        else row = 4 * (lane >> 4) + r;
        const int m = m0 + wm0 + i * MF + row;
        const int n = n0 + wn0 + j * MF + (lane % MF);
        ok = m < p.M && n < p.Cout;
        return (size_t)min(m, p.M - 1) * p.Cout + min(n, p.Cout - 1);
    };
    float sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = min(n0 + wn0 + j * MF + (lane % MF), p.Cout - 1);
        sc[j] = p.scale ? p.scale[n] : 1.f;
        sh[j] = p.shift ? p.shift[n] : 0.f;
    }
    float resq[TM][TN][NACC], agq[TM][TN][NACC];
    const bool has_res = p.residual != nullptr, has_ag = p.actgrad_src != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) { resq[i][j][r] = 0.f; agq[i][j][r] = 1.f; }
    if (has_res) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) { bool ok; resq[i][j][r] = p.residual[element(i, j, r, ok)]; }
    }
    if (has_ag) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) { bool ok; agq[i][j][r] = p.actgrad_src[element(i, j, r, ok)]; }
    }
    auto values = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const float v = acc[i][j][r] * sc[j] + sh[j] + resq[i][j][r];
                    acc[i][j][r] = ACT < 0 ? apply_act(v, p.act) : apply_act(v, ACT);
                }
    };
    if (p.act == CLSLAM_ACT_RELU) values(std::integral_constant<int, CLSLAM_ACT_RELU>{});
    else if (p.act == CLSLAM_ACT_NONE) values(std::integral_constant<int, CLSLAM_ACT_NONE>{});
    else if (p.act == CLSLAM_ACT_ELU) values(std::integral_constant<int, CLSLAM_ACT_ELU>{});
    else values(std::integral_constant<int, -1>{});
    if (has_ag) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) acc[i][j][r] *= act_grad_from_output(agq[i][j][r], p.actgrad_kind);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                bool ok;
                const size_t o = element(i, j, r, ok);
                if (ok) p.out[o] = acc[i][j][r];
            }
}

template <int BM, int BN, int BK, int MF, int WGM>
static int launch_conv(ConvK k, hipStream_t stream) {
    k.tilesM = cdiv(k.M, BM);
    k.tilesN = cdiv(k.Cout, BN);
    k.nblk = k.tilesM * k.tilesN;
    conv_launch(conv_igemm_kernel<BM, BN, BK, MF, WGM>, k.nblk, stream, k);
    return check_launch("conv_igemm");
}

}  // namespace clslam

namespace clslam { int conv3x3_patch_dispatch(const clslam_conv_desc* d, int cfg, hipStream_t stream);
                   int conv3x3_sk_dispatch(const clslam_conv_desc* d, int cfg, hipStream_t stream);
                   int conv3x3_wino_dispatch(const clslam_conv_desc* d, hipStream_t stream);
                   int conv3x3_wino_supported(const clslam_conv_desc* d);
                   int conv3x3_wino_units_per_group(const clslam_conv_desc* d); }

using namespace clslam;

// Tile configurations (BM x BN x BK, MFMA shape, wave grid).  Chosen per layer by clslam_conv2d
// when desc->config < 0; tests sweep them explicitly.
//   0: 128x64x32  32x32x2  2x2 waves (2x1 tiles/wave)   large-M layers with Cout >= 64
//   1:  64x64x32  32x32x2  2x2                           mid layers
//   2:  32x32x32  16x16x4  2x2                           small-M layers (layer3/4, pose decoder)
//   3:  64x32x32  16x16x4  2x2 (2x1 tiles/wave)          Cout == 32
//   4: 128x16x16  16x16x4  4x1 (2x1 tiles/wave)          Cout == 16, Cin % 32 != 0
//   5:  64x32x16  16x16x4  2x2                           BK = 16 fallback
//   6: 128x16x32  16x16x4  4x1                           Cout == 16, Cin % 32 == 0
//  10-13: LDS-patch kernel for 3x3 stride-1 convs (conv_patch.hip): 8x16 px x {64,32,16} ch, 4x16 px x 64 ch
extern "C" int clslam_conv2d_pick_config(const clslam_conv_desc* d) {
    const int Cin = d->ch_a + d->ch_b;
    const int M = d->batch * d->out_h * d->out_w;
    const bool bk32 = (Cin % 32 == 0) && (d->ch_b == 0 || d->ch_a % 32 == 0);
    // Winograd F(2x2,3x3) (conv_wino.hip, config 40) wherever the caller supplied the transformed filter: 2.25x fewer MFMAs
    // (config 40), where a persistent workgroup gets enough (tile, stage) units to amortise the kernel's
    // fixed costs, or the direct kernels are at their weakest (the 6x20 layers): see conv3x3_wino_units_per_group
    if (d->config != -2 && d->workspace != nullptr && conv3x3_wino_supported(d) && Cin >= 64 && d->ch_out >= 64 && !getenv("CLSLAM_NO_WINOGRAD")) {
        static const int min_units = getenv("CLSLAM_WINO_MIN_UNITS") ? atoi(getenv("CLSLAM_WINO_MIN_UNITS")) : 8;
        const int upg = conv3x3_wino_units_per_group(d);
        if (upg >= min_units || (d->out_w <= 24 && upg >= 5)) return 40;
    }
    // 3x3 stride-1: the LDS-patch kernel (conv_patch.hip).  Measured on MI355X (tools/bench_conv.py,
    // B=5 @192x640): 128 px x 16 ch tiles reach 80-104 TFLOP/s on the >= 48x160 layers, 64 px x 16 ch
    // tiles 65-95 TFLOP/s on the smaller ones, 64-px row-major runs 46-70 TFLOP/s on the 6x20 layers
    // (a 4x16 rectangle wastes half its lanes there); all beat every conv_igemm tiling (26-67).
    // Deep, small-M layers (the 6x20 / 12x40 stages, 256-512 channels): the persistent stream-K kernel (conv_sk.hip).
    // Measured on MI355X, B=5 / 2B=10 (profiles/r02_conv_microbench.txt): layer4 51 -> 67, layer4 (pose) 70 -> 85,
    // layer3 (pose) 80 -> 87, upconv_4_0 36 -> 50, pose decoder 46 -> 55, upconv_4_1 79 -> 86, layer4.0 (stride 2)
    // 32 -> 38 TFLOP/s.  On the short-K layers (64-128 channels) every workgroup starts and ends at the same moment and
    // the synchronized first-load / last-store bursts cost more than the even split gains: those stay on the tiled kernel.
    const bool sk_ok = d->workspace != nullptr && d->ksize == 3 && d->ch_out >= 64 && !getenv("CLSLAM_NO_STREAMK") &&
                       d->config != -2;          // config -2: the tiled kernels only (fallback of clslam_conv2d below)
    static const int sk_all = getenv("CLSLAM_SK_ALL") ? atoi(getenv("CLSLAM_SK_ALL")) : 0;   // experiment knob
    if (sk_ok && sk_all && Cin >= sk_all) return d->stride == 2 ? (d->out_w <= 44 ? 31 : 30) : (d->out_w <= 44 ? 32 : 30);
    // Measured per layer shape at B = 5 and B = 1 (tools/bench_conv.py, profiles/r02d_conv_microbench*.txt).  The choice
    // between 128- and 64-pixel tiles is the number of (tile, chunk) units: fewer than five per workgroup and the smaller
    // tile's finer cut wins.  Since the owner of a tile fetches its contributors' slabs four at a time the even split also
    // wins when few tiles exist (B = 1: 14.6 -> 34.5 TFLOP/s on layer4).  A stream-K launch holds every CU (two 220-VGPR
    // waves per SIMD, 120 KB of LDS), so the other two streams of the step stand still behind it; while the B <= 3 steps were
    // bound by the host's launch path that cost more than the faster kernels gained (1.60-1.76 vs 1.55 ms at B = 1) and a
    // 32-tile minimum kept stream-K off there.  With the launch path trimmed the balance flipped (B = 1: 1.29-1.33 vs
    // 1.31-1.33 ms, B = 2 / 3 / 4: -2.8 / -2.7 / -1.7 %, B = 5: -0.4 %): no minimum any more (CLSLAM_SK_FILL=<tiles> restores one).
    static const int sk_fill_min = getenv("CLSLAM_SK_FILL") ? atoi(getenv("CLSLAM_SK_FILL")) : 0;
    const bool sk_fill = (long long)d->batch * cdiv(d->out_h * d->out_w, 128) * cdiv(d->ch_out, 64) >= sk_fill_min;
    // ... and only where the pixel tiles are reasonably full: a 64-pixel run on a 2x4 image (the 64x128 test frames) is
    // 12 % pixels and 88 % padding MFMAs (6x20 in 8x16 tiles, 47 %, still wins: 39.6 vs 32.2 TFLOP/s)
    const int px = d->out_h * d->out_w;
    auto full_enough = [&](int cfg) {
        const int covered = cfg == 32 ? cdiv(px, 128) * 128 : cfg == 33 ? cdiv(px, 64) * 64
                                      : cdiv(d->out_h, cfg == 30 ? 8 : 4) * (cfg == 30 ? 8 : 4) * cdiv(d->out_w, 16) * 16;
        return px * 10 >= covered * 4;
    };
    int sk = 0;
    if (sk_ok && sk_fill && d->stride == 1 && d->out_h == d->in_h + 2 * d->pad - 2 && d->out_w == d->in_w + 2 * d->pad - 2) {
        const long long units128 = (long long)d->batch * cdiv(px, 128) * cdiv(d->ch_out, 64) * (Cin / 16);
        // 256 -> 256 @12x40 at B = 5 is a tie stand-alone (74.0 tiled / 73.6) and 0.5 % slower inside the step: tiled
        const bool tie_case = Cin < 512 && d->ch_out >= 256 && d->out_w > 24 && units128 >= 1280 && M < 4000;
        if (d->out_w <= 44 && Cin >= 256 && !tie_case) sk = units128 >= 1280 ? 32 : 33;
        else if (d->out_w <= 84 && d->out_w > 44 && Cin >= 256 && M >= 4000) sk = 30;
    }
    if (sk_ok && sk_fill && d->stride == 2 && d->out_w <= 24 && Cin >= 256) {
        // the last stage entry (256 -> 512, 12x40 -> 6x20): 8x16 rectangles cover a 6x20 image with 8x32 pixels (47 % real);
        // one 128-pixel stride-2 RUN covers it with 94 % (config 32 with stride 2) when its 13 x 41 input band fits the stage.
        // Measured (MI355X, tools/bench_conv.py): B = 5: 45.0 vs 37.1 TFLOP/s, 2B = 10: 66.1 vs 42.6, step 3.32 -> 3.27 ms;
        // a single triplet has too few (tile, chunk) units for 128-pixel tiles (13.7 vs 19.3 for the 4x16 rectangles, config 31).
        const int spanned = std::min(d->out_h, (127 + d->out_w - 1) / d->out_w + 1);
        const bool band_fits = ((spanned - 1) * 2 + 3) * ((d->out_w - 1) * 2 + 3) <= 544;
        const long long units = (long long)d->batch * cdiv(px, 128) * cdiv(d->ch_out, 64) * (Cin / 16);
        if (units >= 512) sk = (band_fits && !getenv("CLSLAM_NO_SK_RUN_S2")) ? 32 : 30;
        else sk = 31;
    }
    // the pose encoder's layer2.0 (64 -> 128, stride 2, 2B images): 74.7 vs 66.2 TFLOP/s; at B = 5 (M = 9600) the tiled kernel wins
    if (sk_ok && sk_fill && d->stride == 2 && d->out_w > 24 && d->out_w <= 84 && M >= 16000) sk = 30;
    if (sk && full_enough(sk)) return sk;
    if (d->ksize == 3 && d->stride == 1 && d->out_h == d->in_h + 2 * d->pad - 2 && d->out_w == d->in_w + 2 * d->pad - 2) {
        if (d->out_w <= 24) return 22;                         // narrow images: run tiles
        // (config 26, 4x8 px x 32 ch tiles without overhang on 12x40, measured 67 vs 69 TFLOP/s for config 21: not picked)
        // 8x16 (config 20) or 4x16 (config 21) pixel tiles.  Round 2 chose by M alone (>= 16000: 20).  Measured per shape in
        // round 3 (tools/bench_conv.py incl. BENCH_DGRAD=1, B = 1 / 5 / 10): what decides is (a) how much of the tile grid is
        // real pixels -- the padded dgrad domains are 14x42, 26x82, 50x162: 4-row tiles cover 50x162 with 88 % against 82 %
        // (+8.6 %) -- and (b) whether the 8x16 grid has enough workgroups for 1280 slots: 600 of them lose 9-12 % to 1200
        // 4x16 ones (128 -> 128 @24x80 at B = 5, 64 -> 32 @48x160, 128 -> 64 @24x80 at 2B), while on the small 14x42 domains at
        // B >= 5 the larger tile wins by 7-11 % at equal coverage.
        static const bool pick_v2 = !getenv("CLSLAM_PICK_V1");
        if (pick_v2 && d->ch_out >= 32) {
            const double cov20 = (double)px / ((double)cdiv(d->out_h, 8) * 8 * cdiv(d->out_w, 16) * 16);
            const double cov21 = (double)px / ((double)cdiv(d->out_h, 4) * 4 * cdiv(d->out_w, 16) * 16);
            const long long nblk20 = (long long)d->batch * cdiv(d->out_h, 8) * cdiv(d->out_w, 16) * cdiv(d->ch_out, 16);
            if (cov21 > 1.04 * cov20) return 21;
            if (M >= 4000) return nblk20 < 1000 ? 21 : 20;
            return nblk20 >= 400 ? 20 : 21;
        }
        return M >= 16000 ? 20 : 21;                           // 20/21/22 = 12/17/18 with conflict-free LDS rows (24x80 at 2B: 91.5 vs 88.4)
    }
    if (d->ksize == 3 && d->stride == 2 && d->out_h == (d->in_h + 2 * d->pad - 3) / 2 + 1 && d->out_w == (d->in_w + 2 * d->pad - 3) / 2 + 1)
        return 23;
    if (d->ch_out % 32 != 0) return bk32 ? 6 : 4;
    if (!bk32) return 5;
    if (d->ch_out == 32) return 3;
    // measured on MI355X (tools/bench_conv.py): the 32x32 / 16x16x4 tiling wins or ties everywhere
    // except the widest-M 64-channel layers, where 64x32 is ~5 % ahead
    if (d->ch_out == 64 && M >= 30000) return 3;
    return 2;
}

extern "C" int clslam_conv2d(const clslam_conv_desc* d, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CLSLAM_REQUIRE(d, "conv2d: null descriptor");
    if (d->batch == 0 || d->out_h * d->out_w == 0) return CLSLAM_OK;   // an empty batch is a no-op (its tensors have no storage)
    CLSLAM_REQUIRE(d->src_a && d->weight && d->out, "conv2d: null pointer");
    const int Cin = d->ch_a + d->ch_b;
    CLSLAM_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv2d: ksize %d unsupported", d->ksize);
    CLSLAM_REQUIRE(Cin % 16 == 0 && d->ch_a % 16 == 0 && d->ch_out % 16 == 0,
                   "conv2d: channels must be multiples of 16 (Ca=%d Cb=%d Cout=%d)", d->ch_a, d->ch_b, d->ch_out);
    CLSLAM_REQUIRE(!d->upsample_a || (d->in_h % 2 == 0 && d->in_w % 2 == 0), "conv2d: upsample needs even dims");
    CLSLAM_REQUIRE(d->pad_mode == CLSLAM_PAD_ZERO || (d->pad < d->in_h && d->pad < d->in_w), "conv2d: reflect pad too large");
    CLSLAM_REQUIRE(d->ch_b == 0 || d->src_b, "conv2d: src_b missing");
    CLSLAM_REQUIRE(d->stride >= 1 && d->out_h == (d->in_h + 2 * d->pad - d->ksize) / d->stride + 1 &&
                       d->out_w == (d->in_w + 2 * d->pad - d->ksize) / d->stride + 1,
                   "conv2d: output size %dx%d does not match input %dx%d, ksize %d, stride %d, pad %d", d->out_h, d->out_w, d->in_h,
                   d->in_w, d->ksize, d->stride, d->pad);
    ConvK k;
    k.src_a = d->src_a; k.src_b = d->src_b; k.wgt = d->weight; k.scale = d->scale; k.shift = d->shift;
    k.residual = d->residual; k.out = d->out; k.actgrad_src = d->actgrad_src; k.actgrad_kind = d->actgrad_kind;
    k.B = d->batch; k.Hi = d->in_h; k.Wi = d->in_w; k.Ca = d->ch_a; k.Cb = d->ch_b;
    k.Ho = d->out_h; k.Wo = d->out_w; k.Cout = d->ch_out;
    k.ksize = d->ksize; k.stride = d->stride; k.pad = d->pad; k.pad_mode = d->pad_mode;
    k.ups = d->upsample_a; k.act = d->act;
    k.M = d->batch * d->out_h * d->out_w;
    k.tilesM = k.tilesN = k.nblk = 0;
    if (k.M == 0) return CLSLAM_OK;
    int cfg = d->config;
    const bool bk32 = (Cin % 32 == 0) && (d->ch_b == 0 || d->ch_a % 32 == 0);
    if (cfg < 0) cfg = clslam_conv2d_pick_config(d);
    if (cfg == 40) {
        const int rc = conv3x3_wino_dispatch(d, stream);
        if (rc == CLSLAM_OK || d->config >= 0) return rc;
        // an automatically picked Winograd launch that does not fit (scratch smaller than 64 KiB + one slab per workgroup): the
        // direct kernels serve it, like an automatically picked stream-K configuration below
        clslam_conv_desc tiled = *d;
        tiled.config = -2;
        cfg = clslam_conv2d_pick_config(&tiled);
    }
    if (cfg >= 30) {
        const int rc = conv3x3_sk_dispatch(d, cfg, stream);
        if (rc == CLSLAM_OK || d->config >= 0) return rc;
        // an automatically picked stream-K configuration that does not fit this geometry (run tiles on a wide
        // image, scratch too small): the tiled kernel serves it
        clslam_conv_desc tiled = *d;
        tiled.config = -2;
        cfg = clslam_conv2d_pick_config(&tiled);
    }
    if (cfg >= 10) return conv3x3_patch_dispatch(d, cfg, stream);
    const bool need32 = (cfg <= 3 || cfg == 6);
    if (need32 && !bk32) { set_error("conv2d: config %d needs channel multiples of 32", cfg); return CLSLAM_ERR_INVALID; }
    if ((cfg == 0 || cfg == 1) && d->ch_out % 32 != 0) { set_error("conv2d: config %d needs Cout %% 32 == 0", cfg); return CLSLAM_ERR_INVALID; }
    switch (cfg) {
        case 0: return launch_conv<128, 64, 32, 32, 2>(k, stream);
        case 1: return launch_conv<64, 64, 32, 32, 2>(k, stream);
        case 2: return launch_conv<32, 32, 32, 16, 2>(k, stream);
        case 3: return launch_conv<64, 32, 32, 16, 2>(k, stream);
        case 4: return launch_conv<128, 16, 16, 16, 4>(k, stream);
        case 5: return launch_conv<64, 32, 16, 16, 2>(k, stream);
        case 6: return launch_conv<128, 16, 32, 16, 4>(k, stream);
        default: set_error("conv2d: unknown config %d", cfg); return CLSLAM_ERR_INVALID;
    }
}
