// K3/K5 v3: 3x3 convolution as a PERSISTENT, evenly split ("stream-K") implicit GEMM fed by LDS-DMA.
//
// Why (profiles/r01m_*): the per-frame layers are small GEMMs (2.8 GF = 18 us at the fp32-MFMA peak).  With one
// block per output tile (conv_patch.hip) either the tiles are tiny -- one 16x16 MFMA tile per wave, two
// ds_read_b128 per four MFMAs, every block paying its own first-chunk latency and epilogue -- or their number does
// not divide over 256 CUs (300 tiles of 64x64 on layer2: 59 % of the chip).  Here
//   * the work of a launch is the list of (tile, 16-channel chunk) units, tile-major; it is cut into G equal
//     contiguous ranges, one per persistent workgroup (G = number of CUs): every CU gets the same number of MFMAs
//     whatever the tile count (even one tile spread over many CUs: the B=1 / 6x20 layers);
//   * a range is walked from its LAST tile to its first.  The head of a tile that continues in the next range is
//     therefore computed FIRST and parked as a raw partial slab (write-through stores + flag); the workgroup that
//     owns the tile's last chunk reaches it at the END of its own range, adds the parked slabs in a fixed order
//     (deterministic) and runs the epilogue.  Producers never wait; a consumer only waits for workgroups of lower
//     index, which were dispatched before it;
//   * tiles are 128 px x 64 ch (or 64 x 64): each of the 8 waves owns 2x2 (1x2) 16x16 MFMA tiles, i.e. four
//     ds_read_b128 per sixteen v_mfma_f32_16x16x4_f32 instead of two per four;
//   * operands reach LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write), double buffered: the DMA of
//     unit k+1 is issued before the MFMAs of unit k and crosses the single barrier per unit (raw s_barrier; hipcc's
//     __syncthreads() would drain vmcnt).  LDS rows are 64 B (16 channels) and contiguous as the DMA requires; the
//     16-byte slot of a row is XOR-swizzled with bit 2 of the row index ON THE SOURCE ADDRESS so that the
//     ds_read_b128 fragment reads (16 consecutive rows x 4 slots) are bank-conflict free without padding.
//     Zero padding, reflection, nearest-2x upsampling and the skip concat are per-lane address arithmetic of the
//     DMA (out-of-image lanes fetch from a zero page).
// Same fused epilogue and operand convention as conv_patch.hip (lane's float4 = four consecutive k-steps).
#include "common.h"

#include <mutex>
#include <unordered_map>

namespace clslam {

__device__ float g_zero_page[64];   // 256 B of zeros: the DMA source of padded / out-of-range rows

constexpr int kSkFlagOffset = 8192;         // u32 index into the caller's zero-filled workspace head (64 KiB)
constexpr int kSkMaxGroups = 4096;
constexpr int kSkSlabOffsetBytes = 64 << 10;
constexpr unsigned kSkSpinLimit = 1u << 22;

struct SkK {
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
    int tilesX, tilesY, tilesN, tiles, NC, G;
    long long units;
    float* slabs;        // [G][BM*BN] raw partial tiles
    unsigned* flags;     // [G], zero on entry, reset by the consumer
    int b_fastest;       // tile order: 1 = the images of one (spatial, channel) tile position are adjacent (see launch_sk)
    unsigned epoch;      // value a producer publishes in this launch (non-zero, different for consecutive eager launches)
    int dbg;             // measurement probes (CLSLAM_SK_DBG): 1 no epilogue, 2 no hand-off, 4 no MFMA, 8 no DMA
};

// patch rows (pixels) reserved per LDS stage for run tiles; stride-2 runs (the 6x20 outputs of the last stage entry: one
// 128-pixel run covers a whole 120-pixel image, 13 x 41 input pixels) need the larger band
constexpr int sk_run_pp(int bm, int s = 1) { return s == 2 ? 544 : (bm <= 64 ? 336 : 416); }

// TH x TW output pixels (RUN: a run of TH*TW row-major pixels of one image), BN output channels, stride S,
// NWM x NWN waves of TM x TN 16x16 MFMA tiles.
template <int TH, int TW, bool RUN, int S, int BN, int NWM, int NWN>
__global__ __launch_bounds__(NWM * NWN * 64) void conv3x3_sk_kernel(SkK p) {
    constexpr int NW = NWM * NWN, NT = NW * 64;
    constexpr int BM = TH * TW;
    constexpr int TM = BM / (16 * NWM), TN = BN / (16 * NWN);
    constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
    constexpr int PP = RUN ? sk_run_pp(BM, S) : (PH * PW + 15) / 16 * 16;
    constexpr int NPP = PP / 16, NWP = 9 * BN / 16;              // DMA pieces (1 KiB = 16 rows) per unit
    constexpr int STAGE = (PP + 9 * BN) * 16;                    // floats per LDS stage
    static_assert(BM % (16 * NWM) == 0 && BN % (16 * NWN) == 0 && TW % 16 == 0, "tile shape");

    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
    __shared__ int s_flag_ok;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int frow = lane & 15, kg = lane >> 4;
    const int Cin = p.Ca + p.Cb;
    const int HA = p.ups ? (p.Hi >> 1) : p.Hi, WA = p.ups ? (p.Wi >> 1) : p.Wi;

#if CLSLAM_DEVICE_BUILD
    const int grp = xcd_remap((int)blockIdx.x, p.G);            // neighbouring ranges share an XCD's L2
#else
    const int grp = (int)blockIdx.x;                            // the emulator runs blocks in index order
#endif
    const long long u0 = (long long)grp * p.units / p.G, u1 = (long long)(grp + 1) * p.units / p.G;
    if (u0 >= u1) return;
    if (tid == 0) s_flag_ok = 1;                                // cleared by a hand-off that timed out (finish)

    // ---- lane constants of the DMA: row inside a piece, swizzled source slot --------------------------------
    const int drow = lane >> 2;
    const int dslot = (lane & 3) ^ (((drow >> 2) & 1) << 1);
    const int wlane_off = drow * 9 * Cin + dslot * 4;             // weight row (n = drow, tap 0) + slot

    // tile index -> (channel tile, spatial tile, image).  Default: the channel tiles of one spatial tile are adjacent (they
    // share the input patch); b_fastest: the IMAGES of one (channel, spatial) tile are adjacent, so the ~5 consecutive tiles
    // an XCD's workgroups walk share one 64-channel weight slab through that XCD's L2 instead of each fetching their own
    // (layers whose weights are larger than their input: the 256/512-channel stages).
    auto decode_tile = [&](int t, int& tn, int& tx, int& ty, int& b) {
        if (p.b_fastest) {
            b = t % p.B; t /= p.B;
            tn = t % p.tilesN; t /= p.tilesN;
            tx = t % p.tilesX; ty = t / p.tilesX;
        } else {
            tn = t % p.tilesN; t /= p.tilesN;
            tx = t % p.tilesX; t /= p.tilesX;
            ty = t % p.tilesY; b = t / p.tilesY;
        }
    };

    // ---- per-tile state of the DMA cursor ------------------------------------------------------------------
    constexpr int MYP = (NPP + NW - 1) / NW, MYW = (NWP + NW - 1) / NW;
    int offA[MYP], offB[MYP];     // element offsets of this lane's patch rows in src_a / src_b, -1: zero page
    int dma_n0 = 0;
    auto dma_setup_tile = [&](int t) {
        int tn, tx, ty, b;
        decode_tile(t, tn, tx, ty, b);
        dma_n0 = tn * BN;
        const int m0 = tx * BM;
        const int oy0 = RUN ? m0 / p.Wo : ty * TH, ox0 = RUN ? 0 : tx * TW;
        const int pw = RUN ? (p.Wo - 1) * S + 3 : PW;
        const int ph = RUN ? ((min(p.Ho * p.Wo, m0 + BM) - 1) / p.Wo - oy0) * S + 3 : PH;
#pragma unroll
        for (int k = 0; k < MYP; ++k) {
            const int piece = wave + k * NW;
            const int row = piece * 16 + drow;
            offA[k] = -1; offB[k] = -1;
            if (piece < NPP && row < ph * pw) {
                const int pr = row / pw, pc = row - pr * pw;
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
                    offA[k] = ((b * HA + sy) * WA + sx) * p.Ca;
                    offB[k] = ((b * p.Hi + iy) * p.Wi + ix) * p.Cb;
                }
            }
        }
    };
    auto dma_issue = [&](int c, float* stage) {
        if (p.dbg & 8) return;
        const int c0 = c * 16;
        const bool from_a = c0 < p.Ca;
#pragma unroll
        for (int k = 0; k < MYP; ++k) {
            const int piece = wave + k * NW;
            if (piece < NPP) {
                const int off = from_a ? offA[k] : offB[k];
                const float* src = off < 0 ? g_zero_page + dslot * 4
                                           : (from_a ? p.src_a + off + c0 : p.src_b + off + (c0 - p.Ca)) + dslot * 4;
                lds_dma16(src, stage + piece * 256);
            }
        }
#pragma unroll
        for (int k = 0; k < MYW; ++k) {
            const int piece = wave + k * NW;
            if (piece < NWP) {
                const int tap = piece / (BN / 16), nb = (piece % (BN / 16)) * 16;
                const bool ok = dma_n0 + nb + drow < p.Cout;
                const float* src = ok ? p.wgt + (size_t)((dma_n0 + nb) * 9 + tap) * Cin + c0 + wlane_off : g_zero_page + dslot * 4;
                lds_dma16(src, stage + PP * 16 + piece * 256);
            }
        }
    };

    // ---- compute-side lane constants ---------------------------------------------------------------------------
    const int bslot = (kg ^ (((frow >> 2) & 1) << 1)) * 4;
    int brow[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) brow[j] = ((wn * TN + j) * 16 + frow) * 16 + bslot;   // + tap * BN * 16

    f32x4 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    // per-tile state of the compute cursor
    int prow[TM];
    int c_pw = PW;
    auto cmp_setup_tile = [&](int t) {
        if constexpr (RUN) {
            int tn, tx, ty, b;
            decode_tile(t, tn, tx, ty, b);
            const int m0 = tx * BM;
            const int oy0 = m0 / p.Wo;
            c_pw = (p.Wo - 1) * S + 3;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mm = min(m0 + (wm * TM + i) * 16 + frow, p.Ho * p.Wo - 1);
                prow[i] = (mm / p.Wo - oy0) * S * c_pw + (mm % p.Wo) * S;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = (wm * TM + i) * 16 + frow;
                prow[i] = (m / TW) * S * PW + (m % TW) * S;
            }
        }
    };

    // D = W . X^T: the MFMA's row operand is the weight fragment (16 output channels), its column operand the pixel
    // fragment, so a lane ends up with FOUR CONSECUTIVE CHANNELS (4*kg + r) of ONE pixel (frow): the epilogue, the
    // residual and the partial slabs are 16-byte accesses.  The fragments of tap+1 are requested before the MFMAs
    // of tap are issued (hipcc left to itself requests them three MFMAs before their use), and the DMA of the next
    // unit goes out between the first fragment reads and the first MFMA (`between`).
    auto load_frags = [&](const float* Ps, const float* Ws, int tap, float4 (&fa)[TM], float4 (&fb)[TN]) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = prow[i] + ky * c_pw + kx;
            fa[i] = *reinterpret_cast<const float4*>(&Ps[r * 16 + ((kg ^ (((r >> 2) & 1) << 1)) << 2)]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4*>(&Ws[tap * BN * 16 + brow[j]]);
    };
    auto compute = [&](const float* stage, auto&& between) {
        const float* Ps = stage;
        const float* Ws = stage + PP * 16;
        float4 fa[2][TM], fb[2][TN];
        if (!(p.dbg & 4)) load_frags(Ps, Ws, 0, fa[0], fb[0]);
        between();
        if (p.dbg & 4) return;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) load_frags(Ps, Ws, tap + 1, fa[(tap + 1) & 1], fb[(tap + 1) & 1]);
            sched_fence();
            const float4(&ca)[TM] = fa[tap & 1];
            const float4(&cb)[TN] = fb[tap & 1];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float xv = t == 0 ? ca[i].x : t == 1 ? ca[i].y : t == 2 ? ca[i].z : ca[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float wv = t == 0 ? cb[j].x : t == 1 ? cb[j].y : t == 2 ? cb[j].z : cb[j].w;
                        acc[i][j] = mfma_16x16x4(wv, xv, acc[i][j]);
                    }
                }
            sched_fence();
        }
    };

    // ---- end of a tile segment: park the partial, or gather + epilogue ------------------------------------------
    bool publish_pending = false;
    auto finish = [&](int t, int c_lo, int c_hi) {
        const bool owner = c_hi == p.NC;
        if ((p.dbg & 2) && (!owner || c_lo > 0)) return;
        if (!owner) {
            float* slab = p.slabs + (size_t)grp * (BM * BN) + (size_t)wave * (TM * TN * 256) + lane * 4;
            mfma_results_settle();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) coherent_store4(slab + (i * TN + j) * 256, acc[i][j]);
            vmem_drain_visible();
            publish_pending = true;  // the flag goes out behind the next vmcnt(0) + barrier this workgroup reaches anyway
            return;
        }
        if (c_lo > 0) {
            // the head of this tile was computed by the preceding ranges (possibly several): add their slabs, nearest
            // range first -- a fixed order for a given launch geometry.  All their flags are polled at once (one lane
            // each) and the slabs are fetched four at a time: one memory round trip per four contributors instead of
            // two per contributor (flag, then slab) -- the owners reach this point together, at the end of the launch.
            const long long tile_first = (long long)t * p.NC;
            int ncon = 0;
            for (int g2 = grp - 1; g2 >= 0 && (long long)(g2 + 1) * p.units / p.G > tile_first; --g2) ++ncon;
            for (int q = tid; q < ncon; q += NT) {
                unsigned spins = 0;
                // wait for THIS launch's epoch: a flag a late producer of an earlier, timed-out launch sets after its consumer
                // gave up carries that launch's epoch and cannot be mistaken for a slab of this one
                while (coherent_load_u32(&p.flags[grp - 1 - q]) != p.epoch && ++spins < kSkSpinLimit) spin_pause();
                if (spins >= kSkSpinLimit) s_flag_ok = 0;
                uncounted_flag_store(&p.flags[grp - 1 - q], 0u);   // zero again for the next launch on this stream
            }
            __syncthreads();
            const float poison = s_flag_ok ? 0.f : __builtin_nanf("");   // a lost hand-off must not pass silently
            const float* slab0 = p.slabs + (size_t)wave * (TM * TN * 256) + lane * 4;
            for (int base = 0; base < ncon; base += 4) {
                const float* sp[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) sp[q] = slab0 + (size_t)(grp - 1 - min(base + q, ncon - 1)) * (BM * BN);
                f32x4 part[4][TM * TN];
                if constexpr (TM * TN == 4) coherent_load4x4_x4(sp[0], sp[1], sp[2], sp[3], part[0], part[1], part[2], part[3]);
                else if constexpr (TM * TN == 2) coherent_load4x2_x4(sp[0], sp[1], sp[2], sp[3], part[0], part[1], part[2], part[3]);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int r = 0; r < TM * TN; r += 2) coherent_load4x2(sp[q] + r * 256, part[q][r], part[q][r + 1]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (base + q < ncon) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[i][j][r] += part[q][i * TN + j][r] + poison;
                    }
            }
        }
        if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) uncounted_flag_store((unsigned*)p.out, 1u); return; }
        // epilogue (scale/shift = folded BatchNorm or bias, residual, activation, activation-gradient mask):
        // this lane's four channels of pixel frow of every MFMA tile, 16-byte accesses
        int tn, tx, ty, b;
        decode_tile(t, tn, tx, ty, b);
        const int n0 = tn * BN, m0 = tx * BM;
        const int oy0 = RUN ? 0 : ty * TH, ox0 = RUN ? 0 : tx * TW;
        size_t opix[TM];
        bool pix_ok[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = (wm * TM + i) * 16 + frow;
            int oy, ox;
            if constexpr (RUN) {
                pix_ok[i] = m0 + m < p.Ho * p.Wo;
                oy = (m0 + m) / p.Wo; ox = (m0 + m) % p.Wo;
            } else {
                oy = oy0 + m / TW; ox = ox0 + m % TW;
                pix_ok[i] = oy < p.Ho && ox < p.Wo;
            }
            opix[i] = (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout;
        }
        // every operand of the tile is requested before the first one is used (one memory round trip per tile, not
        // one per MFMA tile); a residual and an activation-gradient source never occur together
        const float* extra = p.residual ? p.residual : p.actgrad_src;
        const bool has_res = p.residual != nullptr, has_ag = !has_res && p.actgrad_src != nullptr;
        float4 sc[TN], sh[TN], ex[TM][TN];
        bool ch_ok[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + 4 * kg;
            ch_ok[j] = n < p.Cout;            // Cout is a multiple of 16: a 4-channel group is in or out as a whole
            const int nc = ch_ok[j] ? n : 0;
            sc[j] = p.scale ? *reinterpret_cast<const float4*>(p.scale + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
            sh[j] = p.shift ? *reinterpret_cast<const float4*>(p.shift + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                ex[i][j] = (extra && ch_ok[j] && pix_ok[i]) ? *reinterpret_cast<const float4*>(extra + opix[i] + n)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 res[TM][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float4 v = make_float4(acc[i][j][0] * sc[j].x + sh[j].x, acc[i][j][1] * sc[j].y + sh[j].y,
                                       acc[i][j][2] * sc[j].z + sh[j].z, acc[i][j][3] * sc[j].w + sh[j].w);
                const float4 e = ex[i][j];
                v.x += has_res ? e.x : 0.f; v.y += has_res ? e.y : 0.f; v.z += has_res ? e.z : 0.f; v.w += has_res ? e.w : 0.f;
                v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                v.x *= has_ag ? act_grad_from_output(e.x, p.actgrad_kind) : 1.f; v.y *= has_ag ? act_grad_from_output(e.y, p.actgrad_kind) : 1.f;
                v.z *= has_ag ? act_grad_from_output(e.z, p.actgrad_kind) : 1.f; v.w *= has_ag ? act_grad_from_output(e.w, p.actgrad_kind) : 1.f;
                res[i][j] = v;
            }
        // Every load of this tile has been consumed: a vmcnt(0) the compiler CAN see costs nothing here and leaves its
        // wait-count bookkeeping empty, so that it does not drain vmcnt -- and the LDS-DMA in flight -- at the head of
        // the unit loop.  The stores below are asm (uncounted); dma_wait_all() of the next unit covers them.
        vmem_drain_visible();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + 4 * kg;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (ch_ok[j] && pix_ok[i]) uncounted_store4(p.out + opix[i] + n, res[i][j]);
        }
    };

    // ---- the unit stream: tiles descending, chunks ascending inside a tile segment ------------------------------
    const int t_hi = (int)((u1 - 1) / p.NC), t_lo = (int)(u0 / p.NC);
    auto seg_lo = [&](int t) { return (int)(max(u0, (long long)t * p.NC) - (long long)t * p.NC); };
    auto seg_hi = [&](int t) { return (int)(min(u1, (long long)(t + 1) * p.NC) - (long long)t * p.NC); };

    int dt = t_hi, dc = seg_lo(t_hi);                 // DMA cursor (one unit ahead)
    int ct = t_hi, cc = dc, c_lo = dc, c_hi = seg_hi(t_hi);   // compute cursor
    dma_setup_tile(dt);
    cmp_setup_tile(ct);
    dma_issue(dc, lds);
    const int nunits = (int)(u1 - u0);
    for (int k = 0; k < nunits; ++k) {
        dma_wait_all();             // this wave's pieces of unit k have landed (and its slab stores are acknowledged) ...
        wg_barrier_keep_dma();      // ... everybody's have, and nobody still reads the other stage
        if (publish_pending) {      // the partial slab parked during the previous unit is complete in memory: publish it
            if (tid == 0) uncounted_flag_store(&p.flags[grp], p.epoch);
            publish_pending = false;
        }
        compute(lds + (k & 1) * STAGE, [&]() {
            if (k + 1 < nunits) {
                if (++dc >= seg_hi(dt)) { --dt; dc = seg_lo(dt); dma_setup_tile(dt); }
                dma_issue(dc, lds + ((k + 1) & 1) * STAGE);
            }
        });
        if (++cc >= c_hi) {
            finish(ct, c_lo, c_hi);
            if (k + 1 < nunits) {
                --ct; c_lo = seg_lo(ct); c_hi = seg_hi(ct); cc = c_lo;
                cmp_setup_tile(ct);
                zero_acc();
            }
        }
    }
    if (publish_pending) {          // the parked partial was this workgroup's last piece of work
        stores_complete();
        __syncthreads();
        if (tid == 0) uncounted_flag_store(&p.flags[grp], p.epoch);
    }
}

// Number of CUs of the current device (the persistent grid is sized from it, not from a constant).
int sk_device_cus() {
#if CLSLAM_DEVICE_BUILD
    static thread_local int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached_dev = dev; cached_cus = n;
    }
    return cached_cus;
#else
    return 256;
#endif
}

// Per-workspace launch epoch: consecutive launches on a stream publish different flag values (never 0).  Inside a
// hipGraph capture the argument is frozen, i.e. every replay uses the same value -- the consumer's reset to 0 keeps
// that case working exactly like a 0/1 flag.
unsigned sk_next_epoch(const void* workspace) {
    static std::mutex mu;
    static std::unordered_map<const void*, unsigned> epochs;
    std::lock_guard<std::mutex> lock(mu);
    unsigned& e = epochs[workspace];
    e = e == 0xFFFFFFFFu ? 1u : e + 1u;
    return e;
}

template <int TH, int TW, bool RUN, int S, int BN, int NWM, int NWN>
static int launch_sk(SkK k, const clslam_conv_desc* d, hipStream_t stream) {
    constexpr int BM = TH * TW;
    if (RUN) {
        const int spanned = std::min(k.Ho, (BM - 1 + k.Wo - 1) / k.Wo + 1);
        if (((spanned - 1) * S + 3) * ((k.Wo - 1) * S + 3) > sk_run_pp(BM, S)) { set_error("conv2d: image too wide for stream-K run tiles (Wo=%d, stride %d)", k.Wo, S); return CLSLAM_ERR_INVALID; }
    }
    k.tilesX = RUN ? cdiv(k.Ho * k.Wo, BM) : cdiv(k.Wo, TW);
    k.tilesY = RUN ? 1 : cdiv(k.Ho, TH);
    k.tilesN = cdiv(k.Cout, BN);
    k.tiles = k.tilesX * k.tilesY * k.tilesN * k.B;
    k.NC = (k.Ca + k.Cb) / 16;
    k.units = (long long)k.tiles * k.NC;
    // weights larger than the layer's input: order the tiles so that an XCD's L2 is shared through the weight slab
    k.b_fastest = ((size_t)k.Cout * 9 * (k.Ca + k.Cb) > (size_t)k.B * k.Hi * k.Wi * (k.Ca + k.Cb)) ? 1 : 0;
    if (const char* e = getenv("CLSLAM_SK_B_FASTEST")) k.b_fastest = atoi(e);
    // persistent workgroups: as many per CU as their LDS stages allow (two or three 256-thread groups run their
    // DMA-issue / epilogue / hand-off phases against each other's MFMAs; one 512-thread group has the CU to itself)
    constexpr int PHs = (TH - 1) * S + 3, PWs = (TW - 1) * S + 3;
    constexpr int PPs = RUN ? sk_run_pp(BM, S) : (PHs * PWs + 15) / 16 * 16;
    constexpr size_t lds_bytes = (size_t)2 * (PPs + 9 * BN) * 64 + 16;
    constexpr int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)(163840 / lds_bytes), (size_t)(2048 / (NWM * NWN * 64))));
    // (every workgroup must be co-resident: a consumer spins on lower-indexed producers; per_cu is the LDS / thread limit)
    int G = (d->cu_limit > 0 ? std::min(d->cu_limit, sk_device_cus()) : sk_device_cus()) * per_cu;
    if (const char* e = getenv("CLSLAM_SK_GROUPS")) G = std::max(1, atoi(e));
    G = (int)std::min<long long>(std::min(G, kSkMaxGroups), k.units);
    k.G = G;
    const size_t need = (size_t)kSkSlabOffsetBytes + (size_t)G * BM * BN * sizeof(float);
    if (!d->workspace || d->workspace_bytes < need) {
        set_error("conv2d: the stream-K kernel needs a zero-filled workspace of %zu bytes on the launching stream", need);
        return CLSLAM_ERR_INVALID;
    }
    k.flags = (unsigned*)d->workspace + kSkFlagOffset;
    k.epoch = sk_next_epoch(d->workspace);
    k.slabs = (float*)((char*)d->workspace + kSkSlabOffsetBytes);
    auto kern = conv3x3_sk_kernel<TH, TW, RUN, S, BN, NWM, NWN>;
#if CLSLAM_DEVICE_BUILD
    hipEvent_t e0, e1;
    if (profile_next_events(&e0, &e1)) {
        hipExtLaunchKernelGGL(kern, dim3(G), dim3(NWM * NWN * 64), 0, stream, e0, e1, 0, k);
        return check_launch("conv3x3_sk");
    }
#endif
    hipLaunchKernelGGL(kern, dim3(G), dim3(NWM * NWN * 64), 0, stream, k);
    return check_launch("conv3x3_sk");
}

// Called by clslam_conv2d for configs 30-39.
int conv3x3_sk_dispatch(const clslam_conv_desc* d, int cfg, hipStream_t stream) {
    const int st = d->stride;
    if (d->ksize != 3 || (st != 1 && st != 2)) { set_error("conv2d: stream-K configs need a 3x3 conv with stride 1 or 2"); return CLSLAM_ERR_INVALID; }
    SkK k;
    k.src_a = d->src_a; k.src_b = d->src_b; k.wgt = d->weight; k.scale = d->scale; k.shift = d->shift;
    k.residual = d->residual; k.actgrad_src = d->actgrad_src; k.out = d->out; k.actgrad_kind = d->actgrad_kind;
    k.B = d->batch; k.Hi = d->in_h; k.Wi = d->in_w; k.Ca = d->ch_a; k.Cb = d->ch_b; k.Ho = d->out_h; k.Wo = d->out_w;
    k.Cout = d->ch_out; k.pad = d->pad; k.pad_mode = d->pad_mode; k.ups = d->upsample_a; k.act = d->act;
    k.tilesX = k.tilesY = k.tilesN = k.tiles = k.NC = k.G = 0; k.units = 0; k.slabs = nullptr; k.flags = nullptr;
    k.epoch = 1u; k.b_fastest = 0;
    k.dbg = 0;
    if (const char* e = getenv("CLSLAM_SK_DBG")) k.dbg = atoi(e);
    const bool s2 = st == 2;
    switch (cfg) {
        case 30: return s2 ? launch_sk<8, 16, false, 2, 64, 4, 2>(k, d, stream) : launch_sk<8, 16, false, 1, 64, 4, 2>(k, d, stream);
        case 31: return s2 ? launch_sk<4, 16, false, 2, 64, 4, 2>(k, d, stream) : launch_sk<4, 16, false, 1, 64, 4, 2>(k, d, stream);
        case 32: return s2 ? launch_sk<8, 16, true, 2, 64, 4, 2>(k, d, stream) : launch_sk<8, 16, true, 1, 64, 4, 2>(k, d, stream);
        case 33: if (s2) break; return launch_sk<4, 16, true, 1, 64, 4, 2>(k, d, stream);
        // 256-thread groups, 32 output channels: two or three groups per CU
        case 34: return s2 ? launch_sk<8, 16, false, 2, 32, 4, 1>(k, d, stream) : launch_sk<8, 16, false, 1, 32, 4, 1>(k, d, stream);
        case 35: return s2 ? launch_sk<4, 16, false, 2, 32, 2, 2>(k, d, stream) : launch_sk<4, 16, false, 1, 32, 2, 2>(k, d, stream);
        case 36: if (s2) break; return launch_sk<4, 16, true, 1, 32, 2, 2>(k, d, stream);
        case 37: if (s2) break; return launch_sk<8, 16, true, 1, 32, 4, 1>(k, d, stream);
        default: break;
    }
    set_error("conv2d: unknown stream-K config %d (stride %d)", cfg, st);
    return CLSLAM_ERR_INVALID;
}

}  // namespace clslam
