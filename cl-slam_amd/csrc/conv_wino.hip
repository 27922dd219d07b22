// K3 v4: 3x3 stride-1 convolution as Winograd F(2x2,3x3) on the fp32 MFMA -- persistent, evenly split ("stream-K"), LDS-DMA fed.
//
// Why (profiles/r04_*): three rounds of launch-shape and main-loop work left every direct fp32 convolution at 0.5-0.65 of the
// fp32-MFMA peak -- `t = 9.5 us + flops / 130 TFLOP/s` per launch, MFMA-busy 0.41-0.52.  The matrix pipe itself is the floor:
// 36 MFMA MACs per output and input channel.  F(2x2,3x3) needs 16 per 2x2 output tile = 2.25x fewer, in plain fp32 (the
// reference's own GPU path, cuDNN, picks Winograd for these layers too).  Measured error against fp64 (numpy restatement,
// 64...512 channels): 3.6e-7...6.9e-7 of max|y| against 2.6e-7...3.3e-7 for the direct fp32 form.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
//   * U = G g G^T is formed ONCE per weight tensor (clslam_wino_weight_transform: double arithmetic, rounded once), laid out as
//     the LDS image of a stage: [64-channel tile][8-input-channel stage][16 positions][64 out channels][8 in channels], so a
//     stage's 32 KB are contiguous and reach LDS as 32 one-KiB DMA pieces.
//   * V = B^T d B is formed IN REGISTERS: a lane owns one 2x2-output tile and four input channels (its MFMA operand slot), reads
//     the 4x4 raw pixels of its tile from the LDS patch (16 ds_read_b128) and applies the 32 additions per channel -- the
//     transformed input never exists in memory (the unfused form would move 4x the activations).  The transform of stage k+1
//     is interleaved, slice by slice, with the MFMAs of stage k.
//   * the 16 positions are 16 independent GEMMs  M_p[cout][tile] += U_p[cout][cin] V_p[cin][tile]  on v_mfma_f32_32x32x2_f32: a
//     wave owns 32 tiles x 32 output channels x 16 positions = 256 accumulator registers, one wave per SIMD, a workgroup of
//     four waves 64 tiles x 64 channels.  Per 8-channel stage and wave: 16 + 16 ds_read_b128, 128 v_add/v_sub, 64 MFMAs
//     (4096 matrix cycles) -- the LDS is ~20 % busy, the matrix pipe is what is left to wait for.
//   * A^T M A is applied in registers at the end of a tile segment (lane-local: the 16 positions of a tile live in one lane),
//     BEFORE the stream-K hand-off: partial slabs are 2x2 outputs, not 16 positions.
//   * work = (region, stage) units cut into G equal ranges exactly like conv_sk.hip (same flags, epochs, fixed-order gather).
//     A region is RB images x RH x RW tiles (<= 64), chosen per layer by the host for coverage: 8x8 tiles on 48x160, 6x10 on
//     24x80 and 12x40, two whole 6x20 images.
// Same operand conventions and fused epilogue as the other convolution kernels (scale/shift, residual, activation).
#include "common.h"

#include <type_traits>
#include <utility>

namespace clslam {

unsigned sk_next_epoch(const void* workspace);   // conv_sk.hip: the hand-off flags of a workspace are shared by both kernels
int sk_device_cus();

__device__ float g_wino_zero_page[64];   // DMA source of padded / out-of-range patch rows

constexpr int kWinoFlagOffset = 8192;          // = kSkFlagOffset
constexpr int kWinoMaxGroups = 4096;
constexpr int kWinoSlabOffsetBytes = 64 << 10;
constexpr unsigned kWinoSpinLimit = 1u << 22;
constexpr int kWinoPP = 352;                   // patch rows (pixels) per 16-channel chunk buffer
constexpr int kWinoPatchFloats = kWinoPP * 16; // 22 KiB
constexpr int kWinoUFloats = 16 * 64 * 8;      // 32 KiB: one stage of U
constexpr int kWinoSlabFloats = 64 * 4 * 64;   // 64 tiles x 2x2 outputs x 64 channels

struct WinoK {
    const float* __restrict__ src;
    const float* __restrict__ u;         // clslam_wino_weight_transform layout
    const float* __restrict__ scale;
    const float* __restrict__ shift;
    const float* __restrict__ residual;
    float* __restrict__ out;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, pad, act;
    int RB, RH, RW;                      // region: RB images x RH x RW tiles of 2x2 outputs
    int regB, regY, regX, tilesN, tiles, NS, G;
    long long units;
    float* slabs;                        // [G][kWinoSlabFloats]
    unsigned* flags;                     // [G]
    int b_fastest;
    unsigned epoch;
    unsigned long long* trace;
};

// measurement probes, compile-time (hipcc -DCLSLAM_WINO_DBG=<bits>: 1 no epilogue, 2 no hand-off, 4 no MFMA, 8 no DMA, 16 no input
// transform): a run-time switch around the MFMAs turns the 256 accumulators into phi webs the register allocator spills
#ifndef CLSLAM_WINO_DBG
#define CLSLAM_WINO_DBG 0
#endif
constexpr int kWinoDbg = CLSLAM_WINO_DBG;
// -DCLSLAM_WINO_TRACE=1: thread 0 of every workgroup stamps s_memtime at its phase boundaries into the workspace behind the slabs
// ([G][64] u64: start, prologue transfers landed, prologue done, then per unit: MFMA loop done, finish done, barrier passed)
#ifndef CLSLAM_WINO_TRACE
#define CLSLAM_WINO_TRACE 0
#endif

template <typename F, int... Ps>
__device__ __forceinline__ void for_positions_impl(F&& f, std::integer_sequence<int, Ps...>) { (f(std::integral_constant<int, Ps>{}), ...); }
template <typename F>
__device__ __forceinline__ void for_positions16(F&& f) { for_positions_impl(f, std::make_integer_sequence<int, 16>{}); }
template <typename F>
__device__ __forceinline__ void for_positions4(F&& f) { for_positions_impl(f, std::make_integer_sequence<int, 4>{}); }
template <typename F>
__device__ __forceinline__ void for_positions2(F&& f) { for_positions_impl(f, std::make_integer_sequence<int, 2>{}); }

__global__ __launch_bounds__(256) void conv3x3_wino_kernel(WinoK p) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * kWinoPatchFloats + 3 * kWinoUFloats];      // 140 KiB
    __shared__ int s_flag_ok;
    float* const Pbuf = lds;
    float* const Ubuf = lds + 2 * kWinoPatchFloats;

    const int tid = threadIdx.x, lane = tid & 63;
#if CLSLAM_DEVICE_BUILD
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: per-wave conditions are scalar branches
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, kg = lane >> 5;
    const unsigned lds0 = lds_addr(lds);                             // LDS byte address of the array: DMA destinations are lds0 + constants
    constexpr unsigned kPOff = 0, kUOff = 2 * kWinoPatchFloats;      // float offsets of the patch / U buffers inside `lds`

#if CLSLAM_DEVICE_BUILD
    const int grp = xcd_remap((int)blockIdx.x, p.G);
#else
    const int grp = (int)blockIdx.x;
#endif
    const long long u0 = (long long)grp * p.units / p.G, u1 = (long long)(grp + 1) * p.units / p.G;
    if (u0 >= u1) return;
    if (tid == 0) s_flag_ok = 1;
#if CLSLAM_WINO_TRACE && CLSLAM_DEVICE_BUILD
    int n_stamp = 0;
    auto stamp = [&]() { if (tid == 0 && n_stamp < 64) p.trace[(size_t)grp * 64 + n_stamp] = __builtin_amdgcn_s_memtime(); ++n_stamp; };
#else
    auto stamp = [&]() {};
#endif
    stamp();

    // ---- this lane's tile inside the region (the same for every region of the launch) -----------------------------
    // ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (+32): the 16 lanes of a group take 16
    // CONSECUTIVE tiles, whose patch rows are spread over the bank quads by the row swizzle below.
    const int q = wm * 32 + (int)((0x73261540u >> (4 * (li >> 2))) & 7u) * 4 + (li & 3);
    const int rtiles = p.RH * p.RW;
    const bool tile_valid = q < p.RB * rtiles;
    const int qq = tile_valid ? q : 0;
    const int t_bl = qq / rtiles, t_ty = (qq - t_bl * rtiles) / p.RW, t_tx = qq - t_bl * rtiles - t_ty * p.RW;
    const int PW = 2 * p.RW + 2, PH = 2 * p.RH + 2;
    const int prow0 = (t_bl * PH + 2 * t_ty) * PW + 2 * t_tx;

    auto decode_tile = [&](int t, int& tn, int& rx, int& ry, int& rb) {
        if (p.b_fastest) {
            rb = t % p.regB; t /= p.regB;
            tn = t % p.tilesN; t /= p.tilesN;
            rx = t % p.regX; ry = t / p.regX;
        } else {
            tn = t % p.tilesN; t /= p.tilesN;
            rx = t % p.regX; t /= p.regX;
            ry = t % p.regY; rb = t / p.regY;
        }
    };

    // ---- DMA: patch chunk (16 channels: 22 pieces of 16 rows x 64 B) and U stage (32 contiguous KiB pieces) ------------
    // a lane's 16 bytes land at piece base + 16 * lane: row = lane >> 2, physical slot = lane & 3 holds the logical 4-channel
    // group (slot ^ ((row >> 2) & 3)) -- the swizzle is applied to the SOURCE address
    constexpr int NPP = kWinoPP / 16, MYP = (NPP + 3) / 4;
    const int drow = lane >> 2;
    const int dls = (lane & 3) ^ ((drow >> 2) & 3);
    int offP[MYP];
    int dma_tile = -1, dma_tn = 0;
    auto dma_setup_tile = [&](int t) {
        int tn, rx, ry, rb;
        decode_tile(t, tn, rx, ry, rb);
        dma_tile = t; dma_tn = tn;
        const int y0 = ry * 2 * p.RH - p.pad, x0 = rx * 2 * p.RW - p.pad;
#pragma unroll
        for (int k = 0; k < MYP; ++k) {
            const int row = (wave + 4 * k) * 16 + launder(drow);      // (recomputed per region: not 18 hoisted VGPRs)
            offP[k] = -1;
            if (wave + 4 * k < NPP && row < p.RB * PH * PW) {
                const int bl = row / (PH * PW), r2 = row - bl * (PH * PW);
                const int Y = r2 / PW, X = r2 - Y * PW;
                const int b = rb * p.RB + bl, iy = y0 + Y, ix = x0 + X;
                if (b < p.B && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) offP[k] = ((b * p.Hi + iy) * p.Wi + ix) * p.Cin;
            }
        }
    };
    const float* const zero_src = g_wino_zero_page + dls * 4;
    auto dma_patch_piece = [&](int k, int chunk, int pbuf) {        // pbuf: patch buffer 0 / 1
        if (kWinoDbg & 8) return;
        if (wave + 4 * k < NPP) {
            const float* src = offP[k] < 0 ? zero_src : p.src + offP[k] + chunk * 16 + dls * 4;
            lds_dma16_at(src, lds, lds0, kPOff + pbuf * kWinoPatchFloats + (wave + 4 * k) * 256);
        }
    };
    // a wave's eight U pieces are CONSECUTIVE (two groups of four transfers that share one address pair and one M0 write)
    auto dma_u_group = [&](int g, int tn, int stage, int ubuf) {      // ubuf: U buffer 0 / 1 / 2
        if (kWinoDbg & 8) return;
        const int piece0 = wave * 8 + 4 * g;
        lds_dma16_x4(p.u + ((size_t)tn * p.NS + stage) * kWinoUFloats + piece0 * 256 + lane * 4, lds, lds0,
                     kUOff + ubuf * kWinoUFloats + piece0 * 256);
    };

    // ---- the unit stream: regions descending, stages ascending inside a region segment ---------------------------------
    struct Cur { int t, s, hi; };
    auto seg_lo = [&](int t) { return (int)(max(u0, (long long)t * p.NS) - (long long)t * p.NS); };
    auto seg_hi = [&](int t) { return (int)(min(u1, (long long)(t + 1) * p.NS) - (long long)t * p.NS); };
    auto advance = [&](Cur& c) {
        if (++c.s >= c.hi) { --c.t; c.s = seg_lo(c.t); c.hi = seg_hi(c.t); }
    };
    const int nunits = (int)(u1 - u0);
    const int t_hi = (int)((u1 - 1) / p.NS);

    // ---- compute-side constants ----------------------------------------------------------------------------------
    const int urow = (wn * 32 + li) * 8 + ((kg ^ ((li >> 3) & 1)) << 2);       // + pos * 512
    int raddr[16];    // float offsets of the 4x4 raw pixels of this lane's tile, channel group kg of half 0 (half 1: ^ 8)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = prow0 + a * PW + b;
            raddr[a * 4 + b] = r * 16 + ((kg ^ ((r >> 2) & 3)) << 2);
        }

    AccFile accf;     // the 16 x (32 x 32) accumulator tiles: a0-a255, named by the MFMA statements (intrin.h)

    float4 V[2][16];      // [parity of the unit][position]: the transformed input of this lane's tile, four channels
    auto load_raw = [&](const float* Ps, int half, float4 (&R)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) R[i] = *reinterpret_cast<const float4*>(&Ps[raddr[i] ^ (half << 3)]);
    };
    auto sub4 = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
    auto add4 = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    // B^T d B in place: column pass over the rows of d (slices 0-3: column b), then row pass (slices 4-7: row i)
    auto transform_slice = [&](float4 (&R)[16], int sl) {
        if (sl < 4) {
            const int b = sl;
            const float4 d0 = R[b], d1 = R[4 + b], d2 = R[8 + b], d3 = R[12 + b];
            R[b] = sub4(d0, d2); R[4 + b] = add4(d1, d2); R[8 + b] = sub4(d2, d1); R[12 + b] = sub4(d1, d3);
            pin4(R[b], R[4 + b], R[8 + b], R[12 + b]);
        } else {
            const int i = (sl - 4) * 4;
            const float4 t0 = R[i], t1 = R[i + 1], t2 = R[i + 2], t3 = R[i + 3];
            R[i] = sub4(t0, t2); R[i + 1] = add4(t1, t2); R[i + 2] = sub4(t2, t1); R[i + 3] = sub4(t1, t3);
            pin4(R[i], R[i + 1], R[i + 2], R[i + 3]);
        }
    };

    // the same in sixteen halves of two float4 operations (8 VALU) each, for the gaps between the MFMAs of a position; `mid` carries
    // the second element of a slice from its first half to its second
    float4 tf_mid;
    auto transform_half = [&](float4 (&R)[16], int hs) __attribute__((always_inline)) {
        const int sl = hs >> 1;
        const int i0 = sl < 4 ? sl : (sl - 4) * 4, st = sl < 4 ? 4 : 1;
        if ((hs & 1) == 0) {
            tf_mid = R[i0 + st];
            R[i0] = sub4(R[i0], R[i0 + 2 * st]); R[i0 + st] = add4(tf_mid, R[i0 + 2 * st]);
            pin2(R[i0], R[i0 + st]);
        } else {
            const float4 d2 = R[i0 + 2 * st];
            R[i0 + 2 * st] = sub4(d2, tf_mid); R[i0 + 3 * st] = sub4(tf_mid, R[i0 + 3 * st]);
            pin2(R[i0 + 2 * st], R[i0 + 3 * st]);
        }
    };

    // ---- end of a region segment: output transform, then park the partial or gather + epilogue ----------------------------
    bool publish_pending = false;
    auto finish = [&](int t, int s_lo, int s_hi) __attribute__((always_inline)) -> bool {
        const bool owner = s_hi == p.NS;
        if ((kWinoDbg & 2) && (!owner || s_lo > 0)) return false;
        acc_settle();
        // contributors of this region (owner only): every range that holds an earlier part of it, nearest first
        int ncon = 0;
        float poison = 0.f;
        if (owner && s_lo > 0) {
            const long long tile_first = (long long)t * p.NS;
            for (int g2 = grp - 1; g2 >= 0 && (long long)(g2 + 1) * p.units / p.G > tile_first; --g2) ++ncon;
            for (int c = tid; c < ncon; c += 256) {
                unsigned spins = 0;
                while (coherent_load_u32(&p.flags[grp - 1 - c]) != p.epoch && ++spins < kWinoSpinLimit) spin_pause();
                if (spins >= kWinoSpinLimit) s_flag_ok = 0;
                uncounted_flag_store(&p.flags[grp - 1 - c], 0u);
            }
            __syncthreads();
            poison = s_flag_ok ? 0.f : __builtin_nanf("");
        }
        int tn, rx, ry, rb;
        decode_tile(t, tn, rx, ry, rb);
        const int b = rb * p.RB + t_bl;
        const int oy = (ry * p.RH + t_ty) * 2, ox = (rx * p.RW + t_tx) * 2;
        const int nbase = tn * 64 + wn * 32 + 4 * kg;
        const bool has_res = p.residual != nullptr;
        // Y = A^T M A, lane-local (the 16 positions of a tile live in one lane), ONE OUTPUT ROW dy AT A TIME so that the live set
        // stays far below the 256 VGPRs (the accumulator file is not the compiler's to spill into): T[j] = M_0j + M_1j + M_2j
        // (dy = 0) or M_1j - M_2j - M_3j (dy = 1), Y[dx=0] = T0 + T1 + T2, Y[dx=1] = T1 - T2 - T3.
        // Epilogue operands (both output rows) are requested BEFORE the accumulators are read and transformed: the memory round
        // trip hides behind ~800 instructions of arithmetic.  All results are formed first, then ONE wait (it also covers this
        // unit's LDS-DMA, issued >= 5 positions ago: free) and the sixteen stores back to back -- nothing waits for a store.
        float4 sc[4], sh[4], ex[4][4];       // ex: residual in, result out; [dy * 2 + dx][j]
        bool ch_ok[4], pix_ok[4];
        size_t opix[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nbase + 8 * j;
            ch_ok[j] = n < p.Cout;
            const int nc = ch_ok[j] ? n : 0;
            sc[j] = (owner && p.scale) ? *reinterpret_cast<const float4*>(p.scale + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
            sh[j] = (owner && p.shift) ? *reinterpret_cast<const float4*>(p.shift + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int yy = oy + (y >> 1), xx = ox + (y & 1);
            pix_ok[y] = tile_valid && b < p.B && yy < p.Ho && xx < p.Wo;
            opix[y] = (((size_t)b * p.Ho + yy) * p.Wo + xx) * p.Cout;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ex[y][j] = (owner && has_res && ch_ok[j] && pix_ok[y]) ? *reinterpret_cast<const float4*>(p.residual + opix[y] + nbase + 8 * j)
                                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for_positions2([&](auto dyt) {
            constexpr int dy = decltype(dyt)::value;
            f32x16 Y[2];
            for_positions4([&](auto jt) {
                constexpr int j = decltype(jt)::value;
                f32x16 T;
                // a tile is cleared for the next segment right behind its last read, THROUGH THE MATRIX PIPE (beside this VALU work)
                if constexpr (dy == 0) { T = (acc_read<j>(accf) + acc_read<4 + j>(accf)) + acc_read<8 + j>(accf); acc_zero<j>(accf); }
                else {
                    T = (acc_read<4 + j>(accf) - acc_read<8 + j>(accf)) - acc_read<12 + j>(accf);
                    acc_zero<4 + j>(accf); acc_zero<8 + j>(accf); acc_zero<12 + j>(accf);
                }
                if constexpr (j == 0) Y[0] = T;
                if constexpr (j == 1) { Y[0] = Y[0] + T; Y[1] = T; }
                if constexpr (j == 2) { Y[0] = Y[0] + T; Y[1] = Y[1] - T; }
                if constexpr (j == 3) Y[1] = Y[1] - T;
            });
            if (!owner) {
                float* slab = p.slabs + (size_t)grp * kWinoSlabFloats + (size_t)wave * 4096 + launder(lane) * 4;
                if constexpr (dy == 0) vmem_drain_visible();      // this unit's LDS-DMA (issued long ago): no closing wait later
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = {Y[dx][4 * j], Y[dx][4 * j + 1], Y[dx][4 * j + 2], Y[dx][4 * j + 3]};
                        coherent_store4(slab + ((dy * 2 + dx) * 4 + j) * 256, v);
                    }
                return;
            }
            if (ncon > 0) {
                // the eight slab rows of this output row, one memory round trip per contributor (a region is rarely shared by
                // more than two ranges), added nearest range first
                const float* slab0 = p.slabs + (size_t)wave * 4096 + launder(lane) * 4 + dy * 2048;
                for (int c = 0; c < ncon; ++c) {
                    f32x4 part[8];
                    coherent_load4x8(slab0 + (size_t)(grp - 1 - c) * kWinoSlabFloats, part);
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) Y[dx][4 * j + r] += part[dx * 4 + j][r] + poison;
                }
            }
            if (kWinoDbg & 1) { if (Y[0][0] == 12345.678f) uncounted_flag_store((unsigned*)p.out, 1u); return; }
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 v = make_float4(Y[dx][4 * j] * sc[j].x + sh[j].x, Y[dx][4 * j + 1] * sc[j].y + sh[j].y,
                                           Y[dx][4 * j + 2] * sc[j].z + sh[j].z, Y[dx][4 * j + 3] * sc[j].w + sh[j].w);
                    const float4 e = ex[dy * 2 + dx][j];
                    v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
                    v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act);
                    ex[dy * 2 + dx][j] = v;
                }
        });
        if (owner && !(kWinoDbg & 1)) {
            // every compiler-visible load has been consumed (a vmcnt(0) it can see keeps it from draining vmcnt elsewhere) and this
            // wave's LDS-DMA of the unit has landed: the closing wait of the unit is not needed any more, the stores stay in flight
            vmem_drain_visible();
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ch_ok[j] && pix_ok[y]) uncounted_store4(p.out + opix[y] + nbase + 8 * j, ex[y][j]);
        }
        // (no drain behind the slab stores: the flag goes out behind the NEXT unit's closing wait -- a full one -- and barrier)
        if (!owner) publish_pending = true;
        return !(kWinoDbg & 1);                // true: this wave's transfers have been waited for, only stores are in flight
    };

    // ---- prologue: patches and U stages of units 0 and 1, the transformed input of unit 0 --------------------------------
    // Pipeline invariant at the barrier that opens unit k: U(k) and U(k+1) have landed (three U buffers: U(k+2) is fetched
    // DURING unit k, a whole unit before it is needed -- an LDS-DMA takes ~1 us from issue to landing, half a unit), the patch of
    // unit k+1 has landed (the raw pixels of unit k+1 are read and transformed during unit k), V holds unit k.
    Cur cc{t_hi, seg_lo(t_hi), seg_hi(t_hi)};       // compute cursor (unit k)
    Cur c1 = cc;                                    // unit k + 1
    if (nunits > 1) advance(c1);
    Cur c2 = c1;                                    // unit k + 2
    int rb_idx = 0;                                 // patch buffer of unit k + 1
    int ucur = 0;                                   // U buffer of unit k (k mod 3)
    {
        dma_setup_tile(cc.t);
#pragma unroll
        for (int k = 0; k < MYP; ++k) dma_patch_piece(k, cc.s >> 1, 0);
        dma_u_group(0, dma_tn, cc.s, 0); dma_u_group(1, dma_tn, cc.s, 0);
        if (nunits > 1) {
            if (c1.t != cc.t || (c1.s >> 1) != (cc.s >> 1)) {
                if (c1.t != dma_tile) dma_setup_tile(c1.t);
#pragma unroll
                for (int k = 0; k < MYP; ++k) dma_patch_piece(k, c1.s >> 1, 1);
                rb_idx = 1;
            }
            int tn1, rx, ry, rb;
            decode_tile(c1.t, tn1, rx, ry, rb);
            dma_u_group(0, tn1, c1.s, 1); dma_u_group(1, tn1, c1.s, 1);
        }
        for_positions16([&](auto pt) { acc_zero<decltype(pt)::value>(accf); });     // behind the last issue, beside the transfers' flight
        // U(1), issued last, may stay in flight: unit 0's closing wait covers it
        if (nunits > 1) dma_wait_keep8(); else dma_wait_all();
        wg_barrier_keep_dma();
        stamp();
        if (!(kWinoDbg & 16)) {
            load_raw(Pbuf, cc.s & 1, V[0]);
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) transform_slice(V[0], sl);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) V[0][i] = make_float4(1.f, 1.f, 1.f, 1.f);
        }
    }

    stamp();

    // One unit: 16 positions x 4 MFMAs.  Behind the MFMAs of position pos go, in program order, one DMA piece of the patch of
    // unit k + 2 (positions 0-5: it has to land by the end of this unit) or of U(k + 2) (positions 6-13: may stay in flight
    // across the closing barrier), and one slice of the input transform of unit k + 1 (positions 2-9) -- issue slots in the
    // shadow of the 256 matrix cycles of a position.
    auto unit = [&](int k) __attribute__((always_inline)) {
        const bool has1 = k + 1 < nunits, has2 = k + 2 < nunits;
        if (has2) { c2 = c1; advance(c2); }
        const bool need_patch = has2 && (c2.t != c1.t || (c2.s >> 1) != (c1.s >> 1));
        if (need_patch && c2.t != dma_tile) dma_setup_tile(c2.t);
        int tn2 = 0;
        if (has2) { int rx, ry, rb; decode_tile(c2.t, tn2, rx, ry, rb); }
        const float* Us = Ubuf + ucur * kWinoUFloats;
        const int unn = ucur == 0 ? 2 : ucur - 1;                           // U buffer of unit k + 2
        const float* Pn = Pbuf + rb_idx * kWinoPatchFloats;
        constexpr bool do_tf = !(kWinoDbg & 16);     // unconditional: behind the last unit it transforms rows nobody uses
        constexpr bool do_mm = !(kWinoDbg & 4);
        if (do_tf) load_raw(Pn, has1 ? (c1.s & 1) : 0, V[1]);
        float4 uf[2];
        uf[0] = *reinterpret_cast<const float4*>(&Us[urow]);
        // The four k-steps of a position chain on one accumulator tile: what is issued BETWEEN them is free (the wave would wait
        // for the previous MFMA anyway), what is issued behind the fourth only overlaps that one MFMA.  Gap 0: the U fragment of the
        // next position + a patch piece of unit k + 2 (positions 0-5); gaps 1 and 3: half a slice of the input transform of unit
        // k + 1 (positions 2-9: the raw pixels requested above have landed by then) or four transfers of U(k + 2) (positions 10,
        // 11); gap 2: V(k + 1) moves into the operand registers of positions that are done (positions 10-15).
        for_positions16([&](auto pt) {
            constexpr int pos = decltype(pt)::value;
            const float4 a = uf[pos & 1], bq = V[0][pos];
            if constexpr (do_mm) acc_mfma<pos, true>(accf, a.x, bq.x);
            if constexpr (pos < 15) uf[(pos + 1) & 1] = *reinterpret_cast<const float4*>(&Us[(pos + 1) * 512 + urow]);
            if constexpr (pos < MYP) { if (need_patch) dma_patch_piece(pos, c2.s >> 1, rb_idx ^ 1); }
            if constexpr (do_mm) acc_mfma<pos, false>(accf, a.y, bq.y);
            if constexpr (pos >= 2 && pos < 10 && do_tf) transform_half(V[1], (pos - 2) * 2);
            if constexpr (pos == 10 || pos == 11) { if (has2) dma_u_group(pos - 10, tn2, c2.s, unn); }
            if constexpr (do_mm) acc_mfma<pos, false>(accf, a.z, bq.z);
            if constexpr (pos >= 10 && pos <= 13) {
                constexpr int i0 = (pos - 10) * 3;
                V[0][i0] = V[1][i0]; V[0][i0 + 1] = V[1][i0 + 1]; V[0][i0 + 2] = V[1][i0 + 2];
                pin2(V[0][i0], V[0][i0 + 1]); pin1(V[0][i0 + 2]);
            }
            if constexpr (pos == 14) { V[0][12] = V[1][12]; V[0][13] = V[1][13]; pin2(V[0][12], V[0][13]); }
            if constexpr (pos == 15) { V[0][14] = V[1][14]; pin1(V[0][14]); }
            if constexpr (do_mm) acc_mfma<pos, false>(accf, a.w, bq.w);
            if constexpr (pos >= 2 && pos < 10 && do_tf) transform_half(V[1], (pos - 2) * 2 + 1);
            sched_fence();
        });
        V[0][15] = V[1][15];
        rb_idx ^= need_patch ? 1 : 0;
        ucur = ucur == 2 ? 0 : ucur + 1;
    };
    for (int k = 0; k < nunits; ++k) {
        const int pn_idx = rb_idx;          // patch buffer of unit k + 1
        unit(k);
        stamp();
        // closing wait: 0 = everything (the last units; the unit AFTER one that parked a partial: the slab stores must be complete
        // before its flag goes out), 1 = all but the eight U(k + 2) transfers, 2 = nothing (finish() has waited for the transfers)
        const bool publish_now = publish_pending;      // a slab parked by an earlier unit
        int closing = (k + 2 >= nunits) ? 0 : 1;
        if (cc.s + 1 >= cc.hi) {
            // End of a region segment.  The next unit's transformed input (64 registers) is NOT kept alive across finish() --
            // its live set plus V would not fit the 256 VGPRs, and the accumulator file is not the compiler's to spill into:
            // it is formed again behind it (~500 exposed cycles per segment of >= 4096 x stages matrix cycles).
            closing = finish(cc.t, seg_lo(cc.t), cc.hi) ? 2 : 0;
            if (k + 1 < nunits && !(kWinoDbg & 16)) {
                load_raw(Pbuf + pn_idx * kWinoPatchFloats, c1.s & 1, V[0]);
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) transform_slice(V[0], sl);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) V[0][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (kWinoDbg & 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) V[0][i] = make_float4(1.f, 1.f, 1.f, 1.f);
        }
        stamp();
        cc = c1; c1 = c2;
        // the patch of unit k + 2 (this wave's pieces) has to be there; the eight U(k + 2) pieces issued after it may stay in flight
        if (publish_now) closing = 0;
        if (closing == 0) dma_wait_all(); else if (closing == 1) dma_wait_keep8();
        wg_barrier_keep_dma();
        if (publish_now) {      // the parked slab is complete in memory (every wave drained its stores before the barrier)
            if (tid == 0) uncounted_flag_store(&p.flags[grp], p.epoch);
            publish_pending = false;
        }
        stamp();
    }
    if (publish_pending) {      // the parked partial was this workgroup's last piece of work
        stores_complete();
        __syncthreads();
        if (tid == 0) uncounted_flag_store(&p.flags[grp], p.epoch);
    }
}

// ---- U = G g G^T, written as the LDS image of the kernel's stages --------------------------------------------------------
__global__ void wino_weight_transform_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int tilesN, int NS) {
    const size_t total = (size_t)tilesN * NS * (kWinoUFloats / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int pslot = (int)(idx & 1);
        const int nl = (int)((idx >> 1) & 63);
        const int pos = (int)((idx >> 7) & 15);
        const size_t blk = idx >> 11;
        const int s = (int)(blk % NS), tn = (int)(blk / NS);
        const int ls = pslot ^ ((nl >> 3) & 1);
        const int n = tn * 64 + nl, i = pos >> 2, j = pos & 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < Cout) {
            float o[4];
            for (int e = 0; e < 4; ++e) {
                const int c = s * 8 + ls * 4 + e;
                double g[3][3];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) g[a][b] = (double)w[((size_t)n * 9 + a * 3 + b) * Cin + c];
                // rows of G: (1,0,0), (1/2,1/2,1/2), (1/2,-1/2,1/2), (0,0,1)
                double r[3];
                for (int b = 0; b < 3; ++b)
                    r[b] = i == 0 ? g[0][b] : i == 1 ? 0.5 * (g[0][b] + g[1][b] + g[2][b]) : i == 2 ? 0.5 * (g[0][b] - g[1][b] + g[2][b]) : g[2][b];
                const double val = j == 0 ? r[0] : j == 1 ? 0.5 * (r[0] + r[1] + r[2]) : j == 2 ? 0.5 * (r[0] - r[1] + r[2]) : r[2];
                o[e] = (float)val;
            }
            v = make_float4(o[0], o[1], o[2], o[3]);
        }
        reinterpret_cast<float4*>(u)[idx] = v;
    }
}

// region shape: the fewest workgroup regions that cover the launch (ties: the smaller patch)
static void wino_pick_region(int B, int Ho, int Wo, int& RB, int& RH, int& RW) {
    const int TH = (Ho + 1) / 2, TW = (Wo + 1) / 2;
    long long best = -1;
    int best_rows = 0;
    for (int rh = 1; rh <= std::min(TH, 64); ++rh)
        for (int rw = 1; rw <= std::min(TW, 64 / rh); ++rw) {
            const int rows1 = (2 * rh + 2) * (2 * rw + 2);
            if (rows1 > kWinoPP) continue;
            int rbmax = 1;
            if (rh == TH && rw == TW) rbmax = std::max(1, std::min(std::min(64 / (rh * rw), kWinoPP / rows1), B));
            for (int rb = 1; rb <= rbmax; ++rb) {
                const long long regions = (long long)cdiv(B, rb) * cdiv(TH, rh) * cdiv(TW, rw);
                const int rows = rb * rows1;
                if (best < 0 || regions < best || (regions == best && rows < best_rows)) { best = regions; best_rows = rows; RB = rb; RH = rh; RW = rw; }
            }
        }
}

int conv3x3_wino_supported(const clslam_conv_desc* d) {
    return d->weight_wino != nullptr && d->ksize == 3 && d->stride == 1 && d->ch_b == 0 && !d->upsample_a && d->pad_mode == CLSLAM_PAD_ZERO &&
           d->ch_a % 16 == 0 && d->ch_out % 16 == 0 && (d->pad == 1 || d->pad == 2) && d->actgrad_src == nullptr &&
           d->out_h == d->in_h + 2 * d->pad - 2 && d->out_w == d->in_w + 2 * d->pad - 2;
}

// Called by clslam_conv2d for config 40.
int conv3x3_wino_dispatch(const clslam_conv_desc* d, hipStream_t stream) {
    if (!conv3x3_wino_supported(d)) {
        set_error("conv2d: the Winograd kernel needs a 3x3 stride-1 zero-padded conv of one source with weight_wino set");
        return CLSLAM_ERR_INVALID;
    }
    WinoK k;
    k.src = d->src_a; k.u = d->weight_wino; k.scale = d->scale; k.shift = d->shift; k.residual = d->residual; k.out = d->out;
    k.B = d->batch; k.Hi = d->in_h; k.Wi = d->in_w; k.Cin = d->ch_a; k.Ho = d->out_h; k.Wo = d->out_w; k.Cout = d->ch_out;
    k.pad = d->pad; k.act = d->act;
    wino_pick_region(k.B, k.Ho, k.Wo, k.RB, k.RH, k.RW);
    k.regB = cdiv(k.B, k.RB); k.regY = cdiv((k.Ho + 1) / 2, k.RH); k.regX = cdiv((k.Wo + 1) / 2, k.RW);
    k.tilesN = cdiv(k.Cout, 64);
    k.tiles = k.regB * k.regY * k.regX * k.tilesN;
    k.NS = k.Cin / 8;
    k.units = (long long)k.tiles * k.NS;
    k.b_fastest = ((size_t)k.Cout * 16 * k.Cin > (size_t)k.B * k.Hi * k.Wi * k.Cin) ? 1 : 0;
    if (const char* e = getenv("CLSLAM_SK_B_FASTEST")) k.b_fastest = atoi(e);
    int G = sk_device_cus();
    if (const char* e = getenv("CLSLAM_SK_GROUPS")) G = std::max(1, atoi(e));
    G = (int)std::min<long long>(std::min(G, kWinoMaxGroups), k.units);
    k.G = G;
    const size_t need = (size_t)kWinoSlabOffsetBytes + (size_t)G * kWinoSlabFloats * sizeof(float) + (CLSLAM_WINO_TRACE ? (size_t)G * 64 * 8 : 0);
    if (!d->workspace || d->workspace_bytes < need) {
        set_error("conv2d: the Winograd kernel needs a zero-filled workspace of %zu bytes on the launching stream", need);
        return CLSLAM_ERR_INVALID;
    }
    k.flags = (unsigned*)d->workspace + kWinoFlagOffset;
    k.epoch = sk_next_epoch(d->workspace);
    k.slabs = (float*)((char*)d->workspace + kWinoSlabOffsetBytes);
    k.trace = (unsigned long long*)((char*)d->workspace + kWinoSlabOffsetBytes + (size_t)G * kWinoSlabFloats * sizeof(float));
#if CLSLAM_DEVICE_BUILD
    hipEvent_t e0, e1;
    if (profile_next_events(&e0, &e1)) {
        hipExtLaunchKernelGGL(conv3x3_wino_kernel, dim3(G), dim3(256), 0, stream, e0, e1, 0, k);
        return check_launch("conv3x3_wino");
    }
#endif
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3(G), dim3(256), 0, stream, k);
    return check_launch("conv3x3_wino");
}

}  // namespace clslam

using namespace clslam;

extern "C" size_t clslam_wino_weight_size(int ch_out, int ch_in) {
    if (ch_out <= 0 || ch_in <= 0 || ch_in % 8) return 0;
    return (size_t)cdiv(ch_out, 64) * (ch_in / 8) * kWinoUFloats;
}

extern "C" int clslam_wino_weight_transform(const float* w, float* u, int ch_out, int ch_in, void* stream) {
    CLSLAM_REQUIRE(w && u, "wino_weight_transform: null pointer");
    CLSLAM_REQUIRE(ch_out > 0 && ch_in > 0 && ch_in % 16 == 0, "wino_weight_transform: ch_in must be a multiple of 16");
    const int tilesN = cdiv(ch_out, 64), NS = ch_in / 8;
    const size_t total = (size_t)tilesN * NS * (kWinoUFloats / 4);
    const int nblk = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(wino_weight_transform_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, w, u, ch_out, ch_in, tilesN, NS);
    return check_launch("wino_weight_transform");
}
