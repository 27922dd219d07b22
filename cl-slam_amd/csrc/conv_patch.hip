// K3/K5 v2: 3x3 stride-1 convolution with an LDS-resident input patch.
//
// conv_fwd.hip re-reads its A tile from L2 for each of the 9 taps (arithmetic intensity ~8 flop/B
// from L2 for a 32x32 tile), which measured 40-70 TFLOP/s on MI355X: L2-bandwidth/latency bound.
// Here a block owns a TH x TW output tile of ONE image and BN output channels.  Per chunk of BK input
// channels it stages ONCE in LDS
//     * the (TH+2) x (TW+2) x BK input patch (halo x1.4 instead of x9; zero/reflect padding, nearest-2x
//       upsampling and skip-concat are resolved while filling it, exactly as in conv_fwd.hip), and
//     * the 9 x BN x BK weight slab,
// and runs all 9 taps x BK/4 MFMA k-steps out of LDS: ~48 flop per byte fetched from L2 for the
// 8x16x64 configuration.  The next chunk is prefetched global->registers while the MFMAs of the current
// chunk run; rows are padded to BK+4 floats so the ds_read_b128 fragment reads of neighbouring pixels
// fall on distinct 16-byte bank slots.  Same operand convention as conv_fwd.hip (lane's float4 = four
// consecutive k-steps), same fused epilogue.  Used for every stride-1 3x3 conv whose image is large
// enough to tile (forward, and dgrad on the padded domain with pad = 2).
#include "common.h"

#include <type_traits>

namespace clslam {

constexpr int kSplitKMaxTiles = 16384;   // counters at the head of the caller's workspace (64 KiB)

struct PatchK {
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
    int pad, pad_mode, ups, act;
    int tilesX, tilesY, tilesN, nblk;
    int n_fastest;   // block order inside an XCD: 1 = the output-channel tiles of one spatial tile are adjacent
    // split-K (small-M layers: too few tiles to fill 256 CUs).  ksplit blocks share an output tile, each
    // reducing a contiguous range of input-channel chunks; they park their raw accumulators in ws
    // [split][tile][BM*BN] and the LAST one to arrive (device-scope counter per tile) sums the ksplit
    // slabs in split order -- deterministic whichever block that is -- and runs the epilogue.
    int ksplit;
    float* ws;
    unsigned* counters;   // [tiles], zero on entry, reset by the finishing block
    size_t ws_bytes;      // host side only
};

// RUN = true ("run tiles", for narrow images such as the 6x20 / 12x40 layers where a 4x16 rectangle
// wastes half its lanes): the tile is a run of BM consecutive row-major output pixels of one image and
// the patch is the full-width band of input rows it touches (runtime PH x (Wo+2), bounded by run_pp(BM)).
constexpr int run_pp(int bm) { return bm <= 64 ? 224 : 320; }

// SK: split-K instantiation (its extra live registers cost the plain kernel a wave of occupancy: 96 -> 100
// VGPRs crosses the 5-waves/SIMD allocation boundary, so the unsplit path is compiled without it)
template <int TH, int TW, int BN, int BK, int MF, int WGM, bool RUN, int LDPAD, int S, bool SK>
__global__ __launch_bounds__(256) void conv3x3_patch_kernel(PatchK p) {
    constexpr int BM = TH * TW;
    constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PP = RUN ? run_pp(BM) : PH * PW;   // S = conv stride
    static_assert(!RUN || S == 1, "run tiles are stride-1 only");
    constexpr int LD = BK + LDPAD;   // +4: rows 16-B aligned; +8 with BK=16 makes the 16x16x4 fragment reads conflict-free
    constexpr int WGN = 4 / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / MF, TN = WN / MF;
    constexpr int KG = 64 / MF, KSTEP = KG * 4;
    constexpr int F4 = BK / 4;
    constexpr int P_SLOTS = PP * F4, W_SLOTS = 9 * BN * F4;
    constexpr int P_IT = (P_SLOTS + 255) / 256, W_IT = (W_SLOTS + 255) / 256;
    constexpr int NACC = MF == 32 ? 16 : 4;
    static_assert(WM % MF == 0 && WN % MF == 0 && BK % KSTEP == 0 && TW % 8 == 0, "tile shape");

    __shared__ __attribute__((aligned(16))) float Ps[PP * LD];
    __shared__ __attribute__((aligned(16))) float Wsm[9 * BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;

    // Consecutive logical ids share an XCD (and its 4 MiB L2).  With small weight tensors the BN-channel
    // tiles of ONE spatial tile are made adjacent, so its input patch is fetched from HBM/MALL once and
    // re-read from L2 by the other tiles; with large weights (layer3/4) the spatial tiles of one channel
    // tile are adjacent instead (the weight slab stays in L2, the small input does anyway).
    int logical = xcd_remap((int)blockIdx.x, p.nblk);
    const int tiles_total = SK ? p.nblk / p.ksplit : p.nblk;
    const int split = SK ? logical / tiles_total : 0;
    if (SK) logical -= split * tiles_total;
    const int tile_id = logical;
    int tn;
    if (p.n_fastest) { tn = logical % p.tilesN; logical /= p.tilesN; }
    const int tx = logical % p.tilesX; logical /= p.tilesX;
    const int ty = logical % p.tilesY; logical /= p.tilesY;
    const int b = logical % p.B;
    if (!p.n_fastest) tn = logical / p.B;
    const int n0 = tn * BN;
    // rectangle tiles: (oy0, ox0) origin, compile-time patch width; run tiles: pixel run [m0, m0+BM)
    const int m0 = tx * BM;                                         // RUN only
    const int oy0 = RUN ? m0 / p.Wo : ty * TH, ox0 = RUN ? 0 : tx * TW;
    const int pw = RUN ? p.Wo + 2 : PW;                             // patch row length
    const int ph = RUN ? (min(p.Ho * p.Wo, m0 + BM) - 1) / p.Wo - oy0 + 3 : PH;
    const int npatch = ph * pw;
    const int Cin = p.Ca + p.Cb;
    const int HA = p.ups ? (p.Hi >> 1) : p.Hi, WA = p.ups ? (p.Wi >> 1) : p.Wi;

    // ---- per-thread patch slots: element offsets of the source pixel in src_a / src_b (-1: zero) ----
    int offA[P_IT], offB[P_IT];
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int f = tid + it * 256;
        const int pp = f / F4;
        offA[it] = -1; offB[it] = -1;
        if (f < P_SLOTS && pp < npatch) {
            const int pr = pp / pw, pc = pp - pr * pw;
            int iy = oy0 * S - p.pad + pr, ix = ox0 * S - p.pad + pc;
            bool ok = true;
            if (p.pad_mode == CLSLAM_PAD_REFLECT) {
                iy = reflect_idx(iy, p.Hi); ix = reflect_idx(ix, p.Wi);
                iy = min(max(iy, 0), p.Hi - 1); ix = min(max(ix, 0), p.Wi - 1);   // overhanging tiles
            } else {
                ok = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            }
            if (ok) {
                const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                offA[it] = ((b * HA + sy) * WA + sx) * p.Ca;
                offB[it] = ((b * p.Hi + iy) * p.Wi + ix) * p.Cb;
            }
        }
    }

    // Prefetch loads are unconditional (out-of-image slots read pixel 0 of the image, out-of-range weight
    // rows read row 0) and the zero mask is applied when the registers are written to LDS, AFTER the MFMA
    // block of the current chunk.  Masking right after the load makes the compiler wait for the prefetch
    // before the MFMAs (s_waitcnt vmcnt(0) ahead of the block): no load/compute overlap.
    bool okP[P_IT], okW[W_IT];
    const float* wptr[W_IT];
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        okP[it] = offA[it] >= 0;
        if (!okP[it]) { offA[it] = 0; offB[it] = 0; }
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int f = tid + it * 256;
        const int c4 = f % F4;
        const int n = (f / F4) % BN;
        const int tap = f / (F4 * BN);
        okW[it] = (f < W_SLOTS) && (n0 + n < p.Cout);
        wptr[it] = p.wgt + ((size_t)(okW[it] ? n0 + n : 0) * 9 + (okW[it] ? tap : 0)) * Cin + c4 * 4;
    }
    float4 rp[P_IT], rw[W_IT];
    auto load_global = [&](int c0) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int f = tid + it * 256;
            const int c = c0 + (f % F4) * 4;
            const float* ptr = c < p.Ca ? p.src_a + offA[it] + c : p.src_b + offB[it] + (c - p.Ca);
            rp[it] = *reinterpret_cast<const float4*>(ptr);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) rw[it] = *reinterpret_cast<const float4*>(wptr[it] + c0);
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int f = tid + it * 256;
            float4 v = rp[it];
            if (!okP[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < P_SLOTS) *reinterpret_cast<float4*>(&Ps[(f / F4) * LD + (f % F4) * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int f = tid + it * 256;
            float4 v = rw[it];
            if (!okW[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < W_SLOTS) *reinterpret_cast<float4*>(&Wsm[(f / F4) * LD + (f % F4) * 4]) = v;
        }
    };

    typedef float accv __attribute__((ext_vector_type(NACC)));
    // A wave with a single MFMA tile would issue one serially dependent accumulator chain (each MFMA
    // waits ~8 extra cycles for the previous result); two interleaved chains keep the pipe full.
    constexpr int NCH = (TM * TN == 1) ? 2 : 1;
    accv acc[TM][TN][NCH];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int h = 0; h < NCH; ++h)
#pragma unroll
                for (int r = 0; r < NACC; ++r) acc[i][j][h][r] = 0.f;

    const int frow = lane % MF, kg = lane / MF;
    // LDS row of this lane's pixel for tap (0,0), per M tile
    int prow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wm0 + i * MF + frow;
        if constexpr (RUN) {
            const int mm = min(m0 + m, p.Ho * p.Wo - 1);
            prow[i] = (mm / p.Wo - oy0) * pw + (mm % p.Wo);
        } else {
            prow[i] = (m / TW) * S * PW + (m % TW) * S;
        }
    }

    const int cps = SK ? (Cin / BK + p.ksplit - 1) / p.ksplit : Cin / BK;   // channel chunks per split
    const int c_begin = SK ? split * cps * BK : 0, c_end = SK ? min(Cin, c_begin + cps * BK) : Cin;
    if (c_begin < c_end) load_global(c_begin);
    // Epilogue operands are fetched now, behind the first chunk: loading them after the main loop puts one
    // or two dependent memory latencies (~1-2 us each) on the tail of every block.
    float scv[TN], shv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = min(n0 + wn0 + j * MF + (lane % MF), p.Cout - 1);
        scv[j] = p.scale ? p.scale[n] : 1.f;
        shv[j] = p.shift ? p.shift[n] : 0.f;
    }
    constexpr bool PRE_RES = (TM * TN * NACC <= 4) && !SK && RUN;   // only where it does not cost a wave of occupancy
    float resv[PRE_RES ? TM * TN * NACC : 1];
    if constexpr (PRE_RES) {
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const int m = wm0 + 4 * (lane >> 4) + r;
            int oy, ox;
            bool ok = true;
            if constexpr (RUN) {
                const int mm = min(m0 + m, p.Ho * p.Wo - 1);
                oy = mm / p.Wo; ox = mm % p.Wo;
            } else {
                oy = min(oy0 + m / TW, p.Ho - 1); ox = min(ox0 + m % TW, p.Wo - 1);
            }
            const int n = min(n0 + wn0 + (lane % MF), p.Cout - 1);
            resv[r] = p.residual ? p.residual[(((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + n] : 0.f;
        }
    }
    for (int c0 = c_begin; c0 < c_end; c0 += BK) {
        __syncthreads();           // previous chunk's MFMAs are done with LDS
        store_lds();
        __syncthreads();
        if (c0 + BK < c_end) load_global(c0 + BK);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int kk = 0; kk < BK / KSTEP; ++kk) {
                float4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *reinterpret_cast<const float4*>(&Ps[(prow[i] + ky * pw + kx) * LD + kk * KSTEP + kg * 4]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *reinterpret_cast<const float4*>(&Wsm[((tap * BN) + wn0 + j * MF + frow) * LD + kk * KSTEP + kg * 4]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float av = t == 0 ? fa[i].x : t == 1 ? fa[i].y : t == 2 ? fa[i].z : fa[i].w;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float bv = t == 0 ? fb[j].x : t == 1 ? fb[j].y : t == 2 ? fb[j].z : fb[j].w;
                            accv& a = acc[i][j][t % NCH];
                            if constexpr (MF == 32) a = mfma_32x32x2(av, bv, a);
                            else a = mfma_16x16x4(av, bv, a);
                        }
                    }
                }
            }
        }
    }

    // ---- split-K: park the partial tile, the last block of the tile gathers all of them -------------
    if constexpr (SK) {
        __shared__ int s_last;
        float* slab = p.ws + ((size_t)split * tiles_total + tile_id) * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    int row;
                    if constexpr (MF == 32) row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    else row = 4 * (lane >> 4) + r;
                    coherent_store(&slab[(wm0 + i * MF + row) * BN + wn0 + j * MF + (lane % MF)],
                                   NCH == 2 ? acc[i][j][0][r] + acc[i][j][NCH - 1][r] : acc[i][j][0][r]);
                }
        stores_complete();      // every lane's slab stores are acknowledged ...
        __syncthreads();        // ... before the block announces itself
        if (tid == 0) s_last = (coherent_inc(&p.counters[tile_id]) == (unsigned)(p.ksplit - 1));
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) coherent_store_u32(&p.counters[tile_id], 0u);   // ready for the next launch on this stream
        const float* gather = p.ws + (size_t)tile_id * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    int row;
                    if constexpr (MF == 32) row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    else row = 4 * (lane >> 4) + r;
                    const int e = (wm0 + i * MF + row) * BN + wn0 + j * MF + (lane % MF);
                    float sum = 0.f;
                    for (int sp = 0; sp < p.ksplit; ++sp) sum += coherent_load(&gather[(size_t)sp * tiles_total * (BM * BN) + e]);
                    acc[i][j][0][r] = sum;
                    if (NCH == 2) acc[i][j][NCH - 1][r] = 0.f;
                }
    }

    // ---- epilogue --------------------------------------------------------------------------------
    // Three straight-line phases: (1) the optional operands (residual, activation-gradient source) of ALL the lane's elements
    // loaded back to back from clamped (always valid) addresses, (2) every value computed with the activation switch OUTSIDE the
    // element loop, (3) all stores back to back.  The element-by-element form (load? - value - load? - store per element, the
    // activation's branches in between) made hipcc put an `s_waitcnt vmcnt(0)` in front of every load and store: eight serial
    // memory round trips per workgroup, ~5 us of its ~11 us life on the 16-channel layers (round 5).
    constexpr int NE = TM * TN * NACC;
    auto element = [&](int i, int j, int r, bool& ok) -> size_t {
        int row;
        if constexpr (MF == 32) row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        else row = 4 * (lane >> 4) + r;
        const int m = wm0 + i * MF + row;
        const int n = n0 + wn0 + j * MF + (lane % MF);
        int oy, ox;
        if constexpr (RUN) {
            ok = (m0 + m < p.Ho * p.Wo) && n < p.Cout;
            const int mm = min(m0 + m, p.Ho * p.Wo - 1);
            oy = mm / p.Wo; ox = mm % p.Wo;
        } else {
            oy = oy0 + m / TW; ox = ox0 + m % TW;
            ok = oy < p.Ho && ox < p.Wo && n < p.Cout;
            oy = min(oy, p.Ho - 1); ox = min(ox, p.Wo - 1);
        }
        return (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + min(n, p.Cout - 1);
    };
    float vals[TM][TN][NACC], resq[TM][TN][NACC], agq[TM][TN][NACC];
    const bool has_res = !PRE_RES && p.residual != nullptr, has_ag = p.actgrad_src != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                if constexpr (PRE_RES) resq[i][j][r] = resv[r];
                else resq[i][j][r] = 0.f;
                agq[i][j][r] = 1.f;
            }
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
                    float v = (NCH == 2 ? acc[i][j][0][r] + acc[i][j][NCH - 1][r] : acc[i][j][0][r]) * scv[j] + shv[j];
                    v += resq[i][j][r];
                    vals[i][j][r] = ACT < 0 ? apply_act(v, p.act) : apply_act(v, ACT);
                }
    };
    if (p.act == CLSLAM_ACT_ELU) values(std::integral_constant<int, CLSLAM_ACT_ELU>{});
    else if (p.act == CLSLAM_ACT_RELU) values(std::integral_constant<int, CLSLAM_ACT_RELU>{});
    else if (p.act == CLSLAM_ACT_NONE) values(std::integral_constant<int, CLSLAM_ACT_NONE>{});
    else values(std::integral_constant<int, -1>{});
    if (has_ag) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) vals[i][j][r] *= act_grad_from_output(agq[i][j][r], p.actgrad_kind);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r) {
                bool ok;
                const size_t o = element(i, j, r, ok);
                if (ok) p.out[o] = vals[i][j][r];
            }
    (void)NE;
}

// SKOK: the configuration has a split-K instantiation (the small-M ones: 20-23)
template <int TH, int TW, int BN, int BK, int MF, int WGM, bool RUN = false, int LDPAD = 4, int S = 1, bool SKOK = false>
static int launch_patch(PatchK k, hipStream_t stream) {
    if (RUN) {
        const int spanned = (TH * TW - 1 + k.Wo - 1) / k.Wo + 1;
        if ((spanned + 2) * (k.Wo + 2) > run_pp(TH * TW)) { set_error("conv2d: image too wide for run tiles (Wo=%d)", k.Wo); return CLSLAM_ERR_INVALID; }
    }
    k.tilesX = RUN ? cdiv(k.Ho * k.Wo, TH * TW) : cdiv(k.Wo, TW);
    k.tilesY = RUN ? 1 : cdiv(k.Ho, TH);
    k.tilesN = cdiv(k.Cout, BN);
    const int tiles = k.tilesX * k.tilesY * k.tilesN * k.B;
    // Split-K is OFF unless CLSLAM_SPLITK=<n> asks for it.  Measured on MI355X, standalone, 3 splits:
    // layer4 512->512 @6x20 57 -> 43 us (B=5), 80 -> 75 us (B=10), upconv_4_0 38 -> 26 us; more splits lose
    // again (slab traffic, per-block prologue) and the 256-channel layers do not gain.  Inside the adapt
    // step, where these convs overlap the other branch's kernels, the step time did not move (275.4 vs
    // 275.1 frames/s), so the heuristic stays conservative; the path is kept for B=1 serving-style use.
    const int chunks = (k.Ca + k.Cb) / BK;
    int ksplit = 1;
    if (const char* e = getenv("CLSLAM_SPLITK")) { if (k.ws) ksplit = std::max(1, std::min(atoi(e), chunks)); }
    if (!SKOK || (size_t)ksplit * tiles * (TH * TW * BN) * sizeof(float) > k.ws_bytes || tiles > kSplitKMaxTiles) ksplit = 1;
    k.ksplit = ksplit;
    k.nblk = tiles * ksplit;
    k.n_fastest = ((size_t)k.Cout * 9 * (k.Ca + k.Cb) * 4 <= (size_t)(2 << 20)) ? 1 : 0;
    if (const char* e = getenv("CLSLAM_N_FASTEST")) k.n_fastest = atoi(e);
    if constexpr (SKOK) {
        if (ksplit > 1) {
            conv_launch(conv3x3_patch_kernel<TH, TW, BN, BK, MF, WGM, RUN, LDPAD, S, true>, k.nblk, stream, k);
            return check_launch("conv3x3_patch(split-K)");
        }
    }
    conv_launch(conv3x3_patch_kernel<TH, TW, BN, BK, MF, WGM, RUN, LDPAD, S, false>, k.nblk, stream, k);
    return check_launch("conv3x3_patch");
}

// Called by clslam_conv2d for configs >= 10.
int conv3x3_patch_dispatch(const clslam_conv_desc* d, int cfg, hipStream_t stream) {
    const int st = d->stride;
    if (d->ksize != 3 || (st != 1 && st != 2) || (st == 2) != (cfg == 23) || ((cfg == 24 || cfg == 25) && (d->ch_a + d->ch_b) % 32 != 0) ||
        (cfg == 26 && d->ch_out % 32 != 0)) {
        set_error("conv2d: patch configs need a 3x3 conv with stride 1 (configs 10-22) or 2 (config 23)");
        return CLSLAM_ERR_INVALID;
    }
    if (d->out_h != (d->in_h + 2 * d->pad - 3) / st + 1 || d->out_w != (d->in_w + 2 * d->pad - 3) / st + 1) {
        set_error("conv2d: inconsistent output size for the patch kernel");
        return CLSLAM_ERR_INVALID;
    }
    PatchK k;
    k.src_a = d->src_a; k.src_b = d->src_b; k.wgt = d->weight; k.scale = d->scale; k.shift = d->shift;
    k.residual = d->residual; k.actgrad_src = d->actgrad_src; k.out = d->out; k.actgrad_kind = d->actgrad_kind;
    k.B = d->batch; k.Hi = d->in_h; k.Wi = d->in_w; k.Ca = d->ch_a; k.Cb = d->ch_b; k.Ho = d->out_h; k.Wo = d->out_w;
    k.Cout = d->ch_out; k.pad = d->pad; k.pad_mode = d->pad_mode; k.ups = d->upsample_a; k.act = d->act;
    k.tilesX = k.tilesY = k.tilesN = k.nblk = 0; k.n_fastest = 0;
    k.ksplit = 1; k.ws = nullptr; k.counters = nullptr; k.ws_bytes = 0;
    if (d->workspace && d->workspace_bytes > (size_t)kSplitKMaxTiles * sizeof(unsigned)) {
        k.counters = (unsigned*)d->workspace;
        k.ws = (float*)((char*)d->workspace + (size_t)kSplitKMaxTiles * sizeof(unsigned));
        k.ws_bytes = d->workspace_bytes - (size_t)kSplitKMaxTiles * sizeof(unsigned);
    }
    switch (cfg) {
        case 10: return launch_patch<8, 16, 64, 16, 32, 2>(k, stream);   // 128 px x 64 ch, 32x32x2
        case 11: return launch_patch<8, 16, 32, 16, 32, 4>(k, stream);   // 128 px x 32 ch
        case 12: return launch_patch<8, 16, 16, 16, 16, 4>(k, stream);   // 128 px x 16 ch, 16x16x4
        case 13: return launch_patch<4, 16, 64, 16, 32, 2>(k, stream);   //  64 px x 64 ch
        case 14: return launch_patch<8, 16, 16, 32, 16, 4>(k, stream);   // 128 px x 16 ch, BK = 32
        case 15: return launch_patch<16, 16, 16, 16, 16, 4>(k, stream);  // 256 px x 16 ch
        case 16: return launch_patch<8, 16, 32, 16, 16, 4>(k, stream);   // 128 px x 32 ch on 16x16x4
        case 17: return launch_patch<4, 16, 16, 16, 16, 4>(k, stream);   //  64 px x 16 ch (small images)
        case 18: return launch_patch<4, 16, 16, 16, 16, 4, true>(k, stream);   // 64-px runs x 16 ch (narrow images)
        case 19: return launch_patch<8, 16, 16, 16, 16, 4, true>(k, stream);   // 128-px runs x 16 ch
        case 20: return launch_patch<8, 16, 16, 16, 16, 4, false, 8, 1, true>(k, stream);  // = 12 with conflict-free rows
        case 21: return launch_patch<4, 16, 16, 16, 16, 4, false, 8, 1, true>(k, stream);  // = 17 with conflict-free rows
        case 22: return launch_patch<4, 16, 16, 16, 16, 4, true, 8, 1, true>(k, stream);   // = 18 with conflict-free rows
        case 23: return launch_patch<4, 16, 16, 16, 16, 4, false, 8, 2, true>(k, stream);  // stride-2 convs (encoder stage entries)
        case 24: return launch_patch<4, 16, 16, 32, 16, 4, true, 8>(k, stream);   // = 22 with BK = 32 (half the chunks: latency-bound layers)
        case 25: return launch_patch<4, 16, 16, 32, 16, 4, false, 8>(k, stream);  // = 21 with BK = 32
        case 26: return launch_patch<4, 8, 32, 16, 16, 2, false, 8>(k, stream);   // 4x8 px x 32 ch: images whose width is a multiple
                                                                                   // of 8 but not 16 (12x40: 4x16 tiles waste 1/6)
        default: set_error("conv2d: unknown patch config %d", cfg); return CLSLAM_ERR_INVALID;
    }
}

}  // namespace clslam
