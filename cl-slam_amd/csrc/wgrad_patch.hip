// K15 wgrad v2: weight gradient of the 3x3 stride-1 convolutions with LDS-resident tiles.
//
//   dW[co][tap][ci] = sum_{pixels p} dZ[p][co] * in(p + tap)[ci]
//
// conv_bwd.hip's conv_wgrad_kernel re-gathers the shifted input for every tap from L2 and spends
// most of its time on gather arithmetic for the 16/32-channel layers (measured 5-15 TFLOP/s there).
// Here a block walks a range of TH x TW output tiles; per tile it stages the dZ tile (BM x COT) and the
// (TH+2) x (TW+2) x CIT input patch in LDS once (padding / upsampling / concat resolved while filling
// it, as in conv_patch.hip) and runs all 9 taps from it.  MFMA v_mfma_f32_16x16x4_f32 with the PIXEL
// index as the reduction dimension: lane group g = lane/16 feeds pixel 4*step+g, a = dZ[p][co],
// b = patch[p + tap][ci]; both are conflict-free ds_read_b32 (row stride = 16 mod 32 banks).
// The four waves split the REDUCTION axis: wave w takes the pixel quads ks = w (mod 4) of every tile and
// accumulates all 9 taps (9 x TI x TJ MFMA tiles in registers, no per-wave tap ownership and hence no
// 3/2/2/2 imbalance or exec-masked MFMAs).  Accumulators stay in registers across the block's whole tile
// range; at the end the four waves' copies are summed in wave order through LDS and written once as
// this split's partial (reduced by clslam_reduce_partials / clslam_reduce_multi, deterministic).
#include "common.h"

namespace clslam {

struct WgPatchK {
    const float* __restrict__ dz;
    const float* __restrict__ src_a;
    const float* __restrict__ src_b;
    float* __restrict__ partial;
    int B, Hi, Wi, Ca, Cb, Ho, Wo, Cout;
    int pad, pad_mode, ups;
    int tilesX, tilesY, ntiles, tiles_per_split, co_tiles, ci_tiles;
};

template <int TH, int TW, int COT, int CIT>
__global__ __launch_bounds__(256) void conv3x3_wgrad_patch_kernel(WgPatchK p) {
    constexpr int BM = TH * TW;
    constexpr int PH = TH + 2, PW = TW + 2, PP = PH * PW;
    constexpr int LDZ = (COT % 32 == 16) ? COT : COT + 16;   // row stride = 16 (mod 32) dwords
    constexpr int LDP = (CIT % 32 == 16) ? CIT : CIT + 16;
    constexpr int TI = COT / 16, TJ = CIT / 16;
    constexpr int Z_F4 = BM * COT / 4, P_F4 = PP * CIT / 4;
    constexpr int Z_IT = (Z_F4 + 255) / 256, P_IT = (P_F4 + 255) / 256;

    // one array: [dZ tile | input patch]; the cross-wave reduction of the epilogue re-uses it from the start
    constexpr int SMEM = (BM * LDZ + PP * LDP) > (4 * TI * TJ * 256) ? (BM * LDZ + PP * LDP) : (4 * TI * TJ * 256);
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const Zs = smem;
    float* const Ps = smem + BM * LDZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cot = blockIdx.x % p.co_tiles, cit = blockIdx.x / p.co_tiles;
    const int split = blockIdx.y;
    const int co0 = cot * COT, ci0 = cit * CIT;
    const int Cin = p.Ca + p.Cb;
    const int HA = p.ups ? (p.Hi >> 1) : p.Hi, WA = p.ups ? (p.Wi >> 1) : p.Wi;
    const bool fromA = ci0 < p.Ca;                      // a CIT tile never straddles the concat boundary
    const int csrc = fromA ? ci0 : ci0 - p.Ca;
    const int Csrc = fromA ? p.Ca : p.Cb;

    static_assert(TW % 4 == 0, "a pixel quad must not straddle two tile rows");
    const float* __restrict__ src = fromA ? p.src_a : p.src_b;   // hoisted: selecting inside the tile loop
                                                                 // re-loads the pointer from the kernarg
    f32x4 acc[9][TI][TJ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][i][j][r] = 0.f;

    const int l16 = lane & 15, kg = lane >> 4;
    const int t_beg = split * p.tiles_per_split;
    const int t_end = min(p.ntiles, t_beg + p.tiles_per_split);

    // Software pipeline: the next tile's dZ tile and input patch are fetched global->registers while the
    // MFMAs of the current tile run; zero masks are applied when the registers go to LDS (masking right
    // after the load would make the compiler wait for the prefetch before the MFMA block).
    float4 rz[Z_IT], rp[P_IT];
    bool okz[Z_IT], okp[P_IT];
    auto load_tile = [&](int tile) {
        int q = tile;
        const int tx = q % p.tilesX; q /= p.tilesX;
        const int ty = q % p.tilesY; q /= p.tilesY;
        const int b = q;
        const int oy0 = ty * TH, ox0 = tx * TW;
#pragma unroll
        for (int it = 0; it < Z_IT; ++it) {
            const int f = tid + it * 256;
            const int m = f / (COT / 4), c4 = f % (COT / 4);
            const int oy = oy0 + m / TW, ox = ox0 + m % TW;
            const bool ok = (f < Z_F4) && oy < p.Ho && ox < p.Wo;
            const size_t o = ok ? (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + co0 + c4 * 4 : 0;
            rz[it] = *reinterpret_cast<const float4*>(p.dz + o);
            okz[it] = ok;
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int f = tid + it * 256;
            const int pp = f / (CIT / 4), c4 = f % (CIT / 4);
            const int pr = pp / PW, pc = pp - pr * PW;
            int iy = oy0 - p.pad + pr, ix = ox0 - p.pad + pc;
            bool ok = f < P_F4;
            if (p.pad_mode == CLSLAM_PAD_REFLECT) {
                iy = reflect_idx(iy, p.Hi); ix = reflect_idx(ix, p.Wi);
                iy = min(max(iy, 0), p.Hi - 1); ix = min(max(ix, 0), p.Wi - 1);
            } else {
                ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            }
            size_t o = 0;
            if (ok) {
                if (fromA) {
                    const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                    o = ((size_t)(b * HA + sy) * WA + sx) * Csrc + csrc + c4 * 4;
                } else {
                    o = ((size_t)(b * p.Hi + iy) * p.Wi + ix) * Csrc + csrc + c4 * 4;
                }
            }
            rp[it] = *reinterpret_cast<const float4*>(src + o);
            okp[it] = ok;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < Z_IT; ++it) {
            const int f = tid + it * 256;
            float4 v = rz[it];
            if (!okz[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < Z_F4) *reinterpret_cast<float4*>(&Zs[(f / (COT / 4)) * LDZ + (f % (COT / 4)) * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int f = tid + it * 256;
            float4 v = rp[it];
            if (!okp[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < P_F4) *reinterpret_cast<float4*>(&Ps[(f / (CIT / 4)) * LDP + (f % (CIT / 4)) * 4]) = v;
        }
    };

    if (t_beg < t_end) load_tile(t_beg);
    for (int tile = t_beg; tile < t_end; ++tile) {
        __syncthreads();   // previous tile's MFMAs are done with LDS
        store_tile();
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);
        // ---- MFMA: reduction over the BM pixels of the tile, 4 pixels per instruction -----------------
#pragma unroll 2
        for (int ks = wave; ks < BM / 4; ks += 4) {
            const int pix = ks * 4 + kg;
            const int prow = (pix / TW) * PW + (pix % TW);
            float a[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = Zs[pix * LDZ + i * 16 + l16];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int off = (tap / 3) * PW + (tap % 3);
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const float bv = Ps[(prow + off) * LDP + j * 16 + l16];
#pragma unroll
                    for (int i = 0; i < TI; ++i) acc[tap][i][j] = mfma_16x16x4(a[i], bv, acc[tap][i][j]);
                }
            }
        }
    }

    // ---- sum the four waves' accumulators (fixed order) and write this split's partial:
    //      partial[split][co][tap][ci] ---------------------------------------------------------------
    float* out = p.partial + (size_t)split * p.Cout * 9 * Cin;
    float* red = smem;   // [wave][TI*TJ][lane][4]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();   // tile buffers free: MFMAs / previous round done
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                *reinterpret_cast<f32x4*>(&red[((wave * TI * TJ + i * TJ + j) * 64 + lane) * 4]) = acc[tap][i][j];
        __syncthreads();
        for (int e = tid; e < TI * TJ * 64; e += 256) {
            const int ij = e / 64, ln = e % 64;
            f32x4 s4 = *reinterpret_cast<const f32x4*>(&red[(ij * 64 + ln) * 4]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&red[((w * TI * TJ + ij) * 64 + ln) * 4]);
                s4 += v;
            }
            const int i = ij / TJ, j = ij % TJ;
            const int ci = ci0 + j * 16 + (ln & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + i * 16 + 4 * (ln >> 4) + r;
                out[((size_t)co * 9 + tap) * Cin + ci] = s4[r];
            }
        }
    }
}

template <int TH, int TW, int COT, int CIT>
static int launch_wgp(WgPatchK k, int splits, hipStream_t stream) {
    k.co_tiles = k.Cout / COT;
    k.ci_tiles = (k.Ca + k.Cb) / CIT;
    hipLaunchKernelGGL((conv3x3_wgrad_patch_kernel<TH, TW, COT, CIT>), dim3(k.co_tiles * k.ci_tiles, splits), dim3(256), 0,
                       stream, k);
    return check_launch("conv3x3_wgrad_patch");
}

static void wgp_tile(const clslam_conv_desc* d, int* cot, int* cit) {
    const int Cin = d->ch_a + d->ch_b;
    const bool c32 = d->ch_out % 32 == 0;
    const bool i32 = Cin % 32 == 0 && (d->ch_b == 0 || d->ch_a % 32 == 0);
    *cot = c32 ? 32 : 16;
    *cit = i32 ? 32 : 16;
    if (*cot == 32 && *cit == 16) *cot = 16;   // instantiated shapes: 16x16, 16x32, 32x32
}

// Pixel tile: 8x16 on the wide images; the two deep stages of the 192x640 / 384x1280 pyramids get tiles that cover them
// exactly -- 4x40 on 12x40 (upconv_4_1, upconv_3_0) and 6x20 on 6x20 (upconv_4_0, pose_0 / pose_1) -- instead of the gather
// kernel (conv_bwd.hip's conv_wgrad_kernel: three shifted copies of the input per 32-pixel chunk, 24 flop per staged byte,
// MFMA-busy 0.32): one patch serves all nine taps (round 5).  32x32 channel tiles only.
// OPT-IN (CLSLAM_WGRAD_DEEP_TILES=1).  Measured on MI355X at B = 5 (tools/bench_conv.py): stand-alone upconv_4_1 60 -> 80 TFLOP/s,
// upconv_3_0 37 -> 43, upconv_4_0 35 -> 41, pose_0/1 37 -> 37, with 4 / 15 / 3 / 5 splits instead of 6 / 19 / 5 / 10 (47 MB fewer
// split partials per step) -- and inside the step 3.075 vs 3.076 ms, the five-step frame 9.93 vs 9.85 ms: two 252-VGPR / 79 KB
// workgroups fill a CU's register file, the data-gradient chain they run beside loses what they gain (the same outcome as
// round 4's weight-tile experiment).
static void wgp_pixel_tile(const clslam_conv_desc* d, int* th, int* tw) {
    int cot, cit;
    wgp_tile(d, &cot, &cit);
    *th = 8; *tw = 16;
    const char* env = getenv("CLSLAM_WGRAD_DEEP_TILES");       // (asked when a layer's plan is made, not per launch)
    const bool deep = env && atoi(env) != 0;
    if (cot == 32 && cit == 32 && deep) {
        if (d->out_w == 40 && d->out_h % 4 == 0) { *th = 4; *tw = 40; }
        else if (d->out_w == 20 && d->out_h == 6) { *th = 6; *tw = 20; }
    }
}

}  // namespace clslam

using namespace clslam;

extern "C" int clslam_wgrad_patch_supported(const clslam_conv_desc* d) {
    if (!(d->ksize == 3 && d->stride == 1 && d->out_h == d->in_h + 2 * d->pad - 2 && d->out_w == d->in_w + 2 * d->pad - 2 &&
          d->ch_out % 16 == 0 && (d->ch_a + d->ch_b) % 16 == 0 && d->ch_a % 16 == 0))
        return 0;
    if (d->out_w > 40) return 1;
    int th, tw;
    wgp_pixel_tile(d, &th, &tw);
    return th != 8;          // narrow images: only where an exact tile exists
}

extern "C" int clslam_wgrad_patch_splits(const clslam_conv_desc* d, int target_blocks) {
    int cot, cit, th, tw;
    wgp_tile(d, &cot, &cit);
    wgp_pixel_tile(d, &th, &tw);
    const int cols = (d->ch_out / cot) * ((d->ch_a + d->ch_b) / cit);
    const int ntiles = d->batch * cdiv(d->out_h, th) * cdiv(d->out_w, tw);
    const int splits = std::max(1, std::min(ntiles, cdiv(target_blocks, cols)));
    const int tps = cdiv(ntiles, splits);
    return cdiv(ntiles, tps);
}

extern "C" int clslam_conv_wgrad_patch(const clslam_conv_desc* d, const float* dz, float* partial, int splits, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CLSLAM_REQUIRE(d && dz && partial && splits >= 1 && clslam_wgrad_patch_supported(d), "conv_wgrad_patch: unsupported conv");
    int cot, cit, th, tw;
    wgp_tile(d, &cot, &cit);
    wgp_pixel_tile(d, &th, &tw);
    WgPatchK k;
    k.dz = dz; k.src_a = d->src_a; k.src_b = d->src_b; k.partial = partial;
    k.B = d->batch; k.Hi = d->in_h; k.Wi = d->in_w; k.Ca = d->ch_a; k.Cb = d->ch_b; k.Ho = d->out_h; k.Wo = d->out_w;
    k.Cout = d->ch_out; k.pad = d->pad; k.pad_mode = d->pad_mode; k.ups = d->upsample_a;
    k.tilesX = cdiv(k.Wo, tw); k.tilesY = cdiv(k.Ho, th);
    k.ntiles = k.B * k.tilesX * k.tilesY;
    k.tiles_per_split = cdiv(k.ntiles, splits);
    k.co_tiles = k.ci_tiles = 0;
    if (th == 4) return launch_wgp<4, 40, 32, 32>(k, splits, stream);
    if (th == 6) return launch_wgp<6, 20, 32, 32>(k, splits, stream);
    if (cot == 32) return launch_wgp<8, 16, 32, 32>(k, splits, stream);
    if (cit == 32) return launch_wgp<8, 16, 16, 32>(k, splits, stream);
    return launch_wgp<8, 16, 16, 16>(k, splits, stream);
}
