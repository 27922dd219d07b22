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
//     stage's 32 KB are contiguous and reach LDS as 32 one-KiB DMA pieces (four per wave: ONE address pair and M0 write).
//   * V = B^T d B is formed IN REGISTERS: a lane owns one 2x2-output tile and two input channels of the stage (its MFMA operand
//     slot), reads the 4x4 raw pixels of its tile from the LDS patch (16 ds_read_b64) and applies the 32 additions per channel --
//     the transformed input never exists in memory (the unfused form would move 4x the activations).
//   * the 16 positions are 16 independent GEMMs  M_p[cout][tile] += U_p[cout][cin] V_p[cin][tile]  on v_mfma_f32_16x16x4_f32;
//     a wave owns 16 tiles x 32 output channels x 16 positions = 128 accumulator registers, a workgroup of EIGHT waves
//     (4 x 2) 64 tiles x 64 channels.
//   * A^T M A is applied in registers at the end of a tile segment (lane-local: the 16 positions of a tile live in one lane),
//     BEFORE the stream-K hand-off: partial slabs are 2x2 outputs, not 16 positions.
//   * work = (region, stage) units cut into G equal ranges exactly like conv_sk.hip (same flags, epochs, fixed-order gather).
//     A region is RB images x RH x RW tiles (<= 64), chosen per layer by the host for coverage: 8x8 tiles on 48x160, 6x10 on
//     24x80 and 12x40, two whole 6x20 images.
// Same operand conventions and fused epilogue as the other convolution kernels (scale/shift, residual, activation).
//
// Two waves per SIMD, and why (round 5, tools/micro/, profiles/r05_wino_*): the first version kept all 16 positions of 32 tiles x
// 32 channels in ONE 512-register wave per SIMD (v_mfma_f32_32x32x2_f32, accumulators pinned in a0-a255 by name because sixteen
// 16-register tuples leave hipcc's allocator no slack).  On gfx950 a wave that has issued a v_mfma_f32_* issues nothing else
// until that MFMA has left the pipe -- 8 / 16 / 32 plain v_add between the MFMAs of ONE wave cost 104 / 147 / 218 cycles per
// MFMA instead of 64, whether consecutive MFMAs share an accumulator or not (tools/micro/mfma_clock.hip), while another
// wave's VALU / LDS / s_nop stream on the same SIMD costs the MFMA wave nothing (64.0 cycles per MFMA,
// tools/micro/mfma_valu_share.hip).  Every ds_read, v_add and DMA issue of the input transform was therefore ADDED to the matrix
// time: 6200-7800 cycles per stage for 4096 cycles of MFMAs, 56-103 TFLOP/s.  With the accumulators split over eight waves one
// wave's transform / LDS reads / DMA issue run while its partner on the SIMD holds the matrix pipe: 60-111 TFLOP/s
// (profiles/r05_wino_microbench.txt), and the kernel needs no inline-asm register file.
#include "common.h"
#include <type_traits>


namespace clslam {

unsigned sk_next_epoch(const void* workspace);   // conv_sk.hip: the hand-off flags of a workspace are shared by both kernels
int sk_device_cus();

__device__ float g_wino_zero_page[64];   // DMA source of padded / out-of-range patch rows

constexpr int kWinoFlagOffset = 8192;          // = kSkFlagOffset
constexpr int kWinoMaxGroups = 1024;          // 8 hand-off flags (one per wave) per workgroup in the 8192-entry flag region
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
    int units;                           // < 2^31 (checked by the dispatcher): 32-bit scalar arithmetic per unit
    float* slabs;                        // [G][kWinoSlabFloats]
    unsigned* flags;                     // [G]
    int b_fastest;
    unsigned epoch;
    unsigned long long* trace;
};

// measurement probes, compile-time (hipcc -DCLSLAM_WINO_DBG=<bits>: 1 no epilogue, 2 no hand-off, 4 no MFMA, 8 no DMA, 16 no input
// transform).  Bit 8 is only good for timing the skeleton: without the DMA nothing writes the LDS arrays and hipcc folds the reads)
#ifndef CLSLAM_WINO_DBG
#define CLSLAM_WINO_DBG 0
#endif
constexpr int kWinoDbg = CLSLAM_WINO_DBG;
// -DCLSLAM_WINO_TRACE=1: thread 0 of every workgroup stamps s_memtime at its phase boundaries into the workspace behind the slabs
// ([G][64] u64: start, prologue done (twice), then per unit: MFMA loop done, finish done, barrier passed)
#ifndef CLSLAM_WINO_TRACE
#define CLSLAM_WINO_TRACE 0
#endif


__global__ __launch_bounds__(512) void conv3x3_wino8_kernel(WinoK p) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * kWinoPatchFloats + 3 * kWinoUFloats];      // 140 KiB
    float* const Pbuf = lds;
    float* const Ubuf = lds + 2 * kWinoPatchFloats;

    const int tid = threadIdx.x, lane = tid & 63;
#if CLSLAM_DEVICE_BUILD
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kg = lane >> 4;
    const unsigned lds0 = lds_addr(lds);
    constexpr unsigned kPOff = 0, kUOff = 2 * kWinoPatchFloats;

#if CLSLAM_DEVICE_BUILD
    const int grp = xcd_remap((int)blockIdx.x, p.G);
#else
    const int grp = (int)blockIdx.x;
#endif
    const int u0 = (int)((long long)grp * p.units / p.G), u1 = (int)((long long)(grp + 1) * p.units / p.G);
    if (u0 >= u1) return;
#if CLSLAM_WINO_TRACE >= 3 && CLSLAM_DEVICE_BUILD
    // per-WAVE stamps ([G][8 waves][64] u64): start, prologue done, then per unit: input transform done, 16 positions done, finish()
    // done, DMA wait done, barrier passed (tools/wino_trace_waves.py)
    int n_stamp = 0;
    auto stamp = [&]() { if (lane == 0 && n_stamp < 64) p.trace[((size_t)grp * 8 + wave) * 64 + n_stamp] = __builtin_amdgcn_s_memtime(); ++n_stamp; };
#elif CLSLAM_WINO_TRACE && CLSLAM_DEVICE_BUILD
    int n_stamp = 0;
    auto stamp = [&]() { if (tid == 0 && n_stamp < 64) p.trace[(size_t)grp * 64 + n_stamp] = __builtin_amdgcn_s_memtime(); ++n_stamp; };
#else
    auto stamp = [&]() {};
#endif
    stamp();

    // this lane's tile inside the region: the 16 lanes of an MFMA column take 16 consecutive tiles
    const int q = wm * 16 + li;
    const int rtiles = p.RH * p.RW;
    const bool tile_valid = q < p.RB * rtiles;
    const int qq = tile_valid ? q : 0;
    const int PW = 2 * p.RW + 2, PH = 2 * p.RH + 2;
    int tile_pos;            // (image, tile row, tile column) of the lane's tile inside the region, ONE register across the unit loop
    int prow0;
    {
        const int t_bl = qq / rtiles, t_ty = (qq - t_bl * rtiles) / p.RW, t_tx = qq - t_bl * rtiles - t_ty * p.RW;
        prow0 = (t_bl * PH + 2 * t_ty) * PW + 2 * t_tx;
        tile_pos = (t_bl << 20) | (t_ty << 10) | t_tx;
    }

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

    // ---- DMA: 22 patch pieces per 16-channel chunk (three per wave), 32 U pieces per stage (ONE group of four per wave) --------
    constexpr int NPP = kWinoPP / 16, MYP = (NPP + 7) / 8;
    const int drow = lane >> 2;
    const int dls = (lane & 3) ^ ((drow >> 2) & 3);
    int offP[MYP];
    int dma_tile = -1, dma_tn = 0;
    // which pixel of the region's patch a lane's DMA row is (image, Y, X packed; -1: beyond the patch) does not depend on the tile:
    // decomposed ONCE (two integer divisions per piece on the VALU, ~250 instructions a tile change cost before)
    int prc[MYP];
#pragma unroll
    for (int k = 0; k < MYP; ++k) {
        const int row = (wave + 8 * k) * 16 + drow;
        prc[k] = -1;
        if (wave + 8 * k < NPP && row < p.RB * PH * PW) {
            const int bl = row / (PH * PW), r2 = row - bl * (PH * PW);
            const int Y = r2 / PW, X = r2 - Y * PW;
            prc[k] = (bl << 20) | (Y << 10) | X;
        }
    }
    auto dma_setup_tile = [&](int t) {
        int tn, rx, ry, rb;
        decode_tile(t, tn, rx, ry, rb);
        dma_tile = t; dma_tn = tn;
        const int y0 = ry * 2 * p.RH - p.pad, x0 = rx * 2 * p.RW - p.pad;
#pragma unroll
        for (int k = 0; k < MYP; ++k) {
            const int c = launder(prc[k]);
            const int b = rb * p.RB + (c >> 20), iy = y0 + ((c >> 10) & 1023), ix = x0 + (c & 1023);
            offP[k] = (c >= 0 && b < p.B && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? ((b * p.Hi + iy) * p.Wi + ix) * p.Cin : -1;
        }
    };
    auto dma_patch_piece = [&](int k, int chunk, int pbuf) {
        if (kWinoDbg & 8) return;
        if (wave + 8 * k < NPP) {
            // (lane-dependent parts laundered: hipcc otherwise keeps 64-bit bases per use in VGPRs -- and spills them)
            const int dl = launder(dls) * 4;
            const float* src = offP[k] < 0 ? g_wino_zero_page + dl : p.src + offP[k] + chunk * 16 + dl;
            lds_dma16_at(src, lds, lds0, kPOff + pbuf * kWinoPatchFloats + (wave + 8 * k) * 256);
        }
    };
    auto dma_u_group = [&](int tn, int stage, int ubuf) {
        if (kWinoDbg & 8) return;
        const int piece0 = wave * 4;
        lds_dma16_x4(p.u + ((size_t)tn * p.NS + stage) * kWinoUFloats + piece0 * 256 + launder(lane) * 4, lds, lds0,
                     kUOff + ubuf * kWinoUFloats + piece0 * 256);
    };

    struct Cur { int t, s, hi; };
    auto seg_lo = [&](int t) { return max(u0, t * p.NS) - t * p.NS; };
    auto seg_hi = [&](int t) { return min(u1, (t + 1) * p.NS) - t * p.NS; };
    auto advance = [&](Cur& c) {
        if (++c.s >= c.hi) { --c.t; c.s = seg_lo(c.t); c.hi = seg_hi(c.t); }
    };
    const int nunits = u1 - u0;
    const int t_hi = (u1 - 1) / p.NS;

    // ---- compute-side constants: this lane's operand slot is input channels 2*kg, 2*kg+1 of a stage -------------------------
    // U of a stage in LDS: [16 positions][4 groups of 16 output channels][4 input-channel pairs][16 channels][2] -- the 32 lanes the
    // LDS serves per pass of a ds_read_b64 (li = 0..15 of two channel pairs) read two contiguous 128-byte rows.  (Rounds 5's first
    // layout, [channel][8 input channels] rows of 32 bytes with the 16-byte halves swizzled, put channels li and li + 4 of a pass on
    // the same banks: SQ_LDS_BANK_CONFLICT = 50 % of the U reads' LDS-active cycles, tools/wino_lds_probe.sh.)
    const int urow = wn * 256 + kg * 32 + li * 2;     // + nt * 128 + pos * 512
    // LDS BYTE offsets of the lane's 4x4 raw pixels inside a patch buffer; the address of a read is (offset ^ half) + buffer: one
    // v_xad_u32 (as float indices it took a v_bitop3 and a v_lshl_add per read)
    unsigned raddr[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = prow0 + a * PW + b;
            raddr[a * 4 + b] = (unsigned)(r * 16 + (((kg >> 1) ^ ((r >> 2) & 3)) << 2) + ((kg & 1) << 1)) * 4u;
        }

    f32x4 acc[16][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][n][r] = 0.f;
    };
    zero_acc();

    auto load_raw = [&](int pbuf, int half, f32x2 (&R)[16]) {
        unsigned base = (unsigned)(pbuf * kWinoPatchFloats) * 4u, hx = (unsigned)half << 5;
        // opaque values, one in a scalar and one in a vector register (a VALU instruction reads one SGPR): v_xad_u32 per read;
        // hipcc folds the masks into a v_bitop3 + v_add per read otherwise
        launder_uniform(hx);
        base = (unsigned)launder((int)base);
#pragma unroll
        for (int i = 0; i < 16; ++i) R[i] = lds_read_f32x2(lds, lds0, (raddr[i] ^ hx) + base);
    };
    // B^T d B in place on packed pairs (v_pk_add_f32: the lane's two input channels per instruction), sixteen halves of two
    // operations: column pass over the rows of d (slices 0-3), row pass (4-7)
    f32x2 tf_mid;
    auto transform_half = [&](f32x2 (&R)[16], int hs) __attribute__((always_inline)) {
        const int sl = hs >> 1;
        const int i0 = sl < 4 ? sl : (sl - 4) * 4, st = sl < 4 ? 4 : 1;
        if ((hs & 1) == 0) {
            tf_mid = R[i0 + st];
            R[i0] = R[i0] - R[i0 + 2 * st]; R[i0 + st] = tf_mid + R[i0 + 2 * st];
        } else {
            const f32x2 d2 = R[i0 + 2 * st];
            R[i0 + 2 * st] = d2 - tf_mid; R[i0 + 3 * st] = tf_mid - R[i0 + 3 * st];
        }
    };

    // ---- end of a region segment ----------------------------------------------------------------------------------------
    // finish() of a partial segment marks the wave's flag for publication: it goes out at the wave's next-but-one DMA wait
    bool publish_pending = false, publish_armed = false;
    auto finish = [&](int t, int s_lo, int s_hi) __attribute__((always_inline)) -> bool {
        const bool owner = s_hi == p.NS;
        if ((kWinoDbg & 2) && (!owner || s_lo > 0)) return false;
        int ncon = 0;
        float poison = 0.f;
        if (owner && s_lo > 0) {
            // Hand-offs are per WAVE: wave w of the owner adds exactly the slab rows wave w of a contributor wrote, so every wave
            // waits for its own eight flags and no workgroup barrier sits inside finish() (the two wave groups of a workgroup
            // reach it half a unit apart).  Flags hold the launch's epoch: nothing has to be reset.
            const int tile_first = t * p.NS;
            for (int g2 = grp - 1; g2 >= 0 && (int)((long long)(g2 + 1) * p.units / p.G) > tile_first; --g2) ++ncon;
            float bad = 0.f;
            for (int c = lane; c < ncon; c += 64) {
                unsigned spins = 0;
                while (coherent_load_u32(&p.flags[(grp - 1 - c) * 8 + wave]) != p.epoch && ++spins < kWinoSpinLimit) spin_pause();
                if (spins >= kWinoSpinLimit) bad = 1.f;
            }
            poison = wave_sum(bad) > 0.f ? __builtin_nanf("") : 0.f;
        }
        if (CLSLAM_WINO_TRACE == 2) stamp();      // flags seen
        int tn, rx, ry, rb;
        decode_tile(t, tn, rx, ry, rb);
        const int tp = launder(tile_pos);
        const int b = rb * p.RB + (tp >> 20);
        const int oy = (ry * p.RH + ((tp >> 10) & 1023)) * 2, ox = (rx * p.RW + (tp & 1023)) * 2;
        const int nbase = tn * 64 + wn * 32 + 4 * kg;            // + 16 * nt
        const bool has_res = p.residual != nullptr;
        // the kernel serves ReLU and identity epilogues only (conv3x3_wino_supported): one v_max per element instead of the five-way
        // activation switch (expm1f and all) compiled 32 times into finish()
        const float act_floor = p.act == CLSLAM_ACT_RELU ? 0.f : -__builtin_inff();
        bool pix_ok[4];
        unsigned opix[4];                  // element offsets fit 32 bits (checked by the dispatcher): four registers, not eight
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int yy = oy + (y >> 1), xx = ox + (y & 1);
            pix_ok[y] = tile_valid && b < p.B && yy < p.Ho && xx < p.Wo;
            opix[y] = (unsigned)(((b * p.Ho + yy) * p.Wo + xx) * p.Cout);
        }
        float* slab = p.slabs + (size_t)grp * kWinoSlabFloats + (size_t)wave * 2048 + launder(lane) * 4;     // rows [n][2x2 output], 1 KiB each
        const float* slab0 = p.slabs + (size_t)wave * 2048 + launder(lane) * 4;
        if (!owner) {
            mfma_results_settle();
            vmem_drain_visible();          // this unit's LDS-DMA (issued long ago): no DMA wait behind finish()
        }
        // The two 16-channel halves of the wave's tile one after the other: with both in flight (round 5) the 128 accumulators, 32
        // residual registers and the partial transforms did not fit 256 VGPRs, and an accumulator reloaded from scratch at the end
        // of finish() made hipcc drain vmcnt(0) -- and the LDS-DMA in flight -- at the third MFMA position of EVERY unit.  The
        // round trips this serialises are covered by the partner wave's MFMAs (schedule below).
        float4 outv[4][2];
        bool ch_ok[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int ch = nbase + 16 * n;
            ch_ok[n] = ch < p.Cout;
            const int nc = ch_ok[n] ? ch : 0;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), ex[4];
            if (owner) {
                if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + nc);
                if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + nc);
            }
#pragma unroll
            for (int y = 0; y < 4; ++y)
                ex[y] = (owner && has_res && ch_ok[n] && pix_ok[y]) ? *reinterpret_cast<const float4*>(p.residual + opix[y] + ch)
                                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            // Y = A^T M A, lane-local: T_0[j] = M_0j + M_1j + M_2j, T_1[j] = M_1j - M_2j - M_3j, Y[dy][0] = T0 + T1 + T2, Y[dy][1] = T1 - T2 - T3
            f32x4 T0[4], T1[4], Y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T0[j] = (acc[j][n] + acc[4 + j][n]) + acc[8 + j][n];
                T1[j] = (acc[4 + j][n] - acc[8 + j][n]) - acc[12 + j][n];
            }
            Y[0] = (T0[0] + T0[1]) + T0[2]; Y[1] = (T0[1] - T0[2]) - T0[3];
            Y[2] = (T1[0] + T1[1]) + T1[2]; Y[3] = (T1[1] - T1[2]) - T1[3];
            if (!owner) {
#pragma unroll
                for (int y = 0; y < 4; ++y) coherent_store4(slab + (n * 4 + y) * 256, Y[y]);
                continue;
            }
            for (int c = 0; c < ncon; ++c) {
                f32x4 part[4];
                coherent_load4x4(slab0 + (size_t)(grp - 1 - c) * kWinoSlabFloats + n * 1024, part[0], part[1], part[2], part[3]);
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Y[y][r] += part[y][r] + poison;
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                float4 v = make_float4(Y[y][0] * sc.x + sh.x, Y[y][1] * sc.y + sh.y, Y[y][2] * sc.z + sh.z, Y[y][3] * sc.w + sh.w);
                v.x += ex[y].x; v.y += ex[y].y; v.z += ex[y].z; v.w += ex[y].w;
                v.x = fmaxf(v.x, act_floor); v.y = fmaxf(v.y, act_floor); v.z = fmaxf(v.z, act_floor); v.w = fmaxf(v.w, act_floor);
                outv[y][n] = v;
            }
        }
        if (!owner) {
            publish_pending = true;        // (the slab stores have been acknowledged by then without a wait of their own)
            return true;
        }
        if (CLSLAM_WINO_TRACE == 2) stamp();      // transformed, gathered
        if (kWinoDbg & 1) { if (outv[0][0].x == 12345.678f) uncounted_flag_store((unsigned*)p.out, 1u); return false; }
        // every compiler-visible load has been consumed and this wave's LDS-DMA of the unit has landed: no DMA wait behind
        // finish(), the stores stay in flight
        vmem_drain_visible();
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                if (ch_ok[n] && pix_ok[y]) uncounted_store4(p.out + opix[y] + nbase + 16 * n, outv[y][n]);
        return true;
    };

    // ---- schedule: one workgroup barrier per unit.  Invariant at the barrier that opens unit k: U(k), U(k+1) and the patch of unit
    // k have landed; U(k+2) and a new patch for unit k+1 are fetched during unit k.
    // What a unit costs (round 6, per-wave s_memtime stamps, profiles/r06_wino_waves_*.txt): its 4096 matrix cycles per SIMD plus
    // everything else its two waves issue -- the SIMD runs one wave's VALU / LDS instructions and its partner's MFMAs mostly one
    // after the other, whatever the phase relation of the two.  Two schedules that tried to hide the input transform were built
    // and measured on the MI355X, both lost, both are gone (profiles/r06_wino_stagger.txt):
    //   * the two waves of a SIMD half a unit apart (two barriers per unit): the transform took 2400-3000 cycles beside the
    //     partner's MFMAs against 1450 when both waves transform together -- the sum stayed, the second barrier cost 15-20 %;
    //   * the raw pixels of unit k+1 read behind the positions of unit k into the operand registers they release: 32 more live
    //     registers at the top of a unit, hipcc spilled the DMA addresses and drained vmcnt(0) between the patch pieces (-10 %).
    // What paid (+13...20 % over round 5 on every layer and batch size) is fewer instructions beside the MFMAs and no drain of the
    // DMA in flight: packed additions (v_pk_add_f32) in the transform, one v_xad_u32 per read address, U fragments requested two
    // positions ahead, the DMA row decomposition hoisted out of the loop, finish() without a workgroup barrier and without a
    // register spill (an accumulator reloaded from scratch at its end made hipcc drain vmcnt(0) at the third position of EVERY
    // unit in round 5; the ISA of this loop has no compiler-placed vmcnt wait any more: tools/isa_scan.py wino8).
    Cur cc{t_hi, seg_lo(t_hi), seg_hi(t_hi)};       // unit k
    Cur c1 = cc;                                    // unit k + 1
    if (nunits > 1) advance(c1);
    Cur c2 = c1;                                    // unit k + 2
    auto new_patch = [](const Cur& a, const Cur& b) { return b.t != a.t || (b.s >> 1) != (a.s >> 1); };
    auto issue_patch = [&](const Cur& c, int pbuf) {
        if (c.t != dma_tile) dma_setup_tile(c.t);
#pragma unroll
        for (int k = 0; k < MYP; ++k) dma_patch_piece(k, c.s >> 1, pbuf);
    };
    auto issue_u = [&](const Cur& c, int ubuf) {
        int tn, rx, ry, rb;
        decode_tile(c.t, tn, rx, ry, rb);
        dma_u_group(tn, c.s, ubuf);
    };
    int pb = 0, ub = 0;                             // patch / U buffer of unit k
    {
        issue_patch(cc, 0);
        issue_u(cc, 0);
        if (nunits > 1) issue_u(c1, 1);
        dma_wait_all();
        wg_barrier_keep_dma();
    }
    stamp();
    if (CLSLAM_WINO_TRACE && CLSLAM_WINO_TRACE < 3) stamp();

    auto publish = [&]() {
        // this wave's slab rows (stored a unit ago) have been acknowledged: its own flag
        dma_wait_all();
        if (lane == 0) uncounted_flag_store(&p.flags[grp * 8 + wave], p.epoch);
        publish_armed = false;
    };
    // U operand fragments are requested TWO positions ahead: the faster wave of a SIMD runs a position in ~130 cycles, less than an
    // LDS round trip beside the DMA traffic
    constexpr int kUAfter = 4;                      // U(k+2) goes out behind this many positions
    auto uf_load = [&](const float* Us, float2 (&uf)[3][2], int pos) __attribute__((always_inline)) {
        uf[pos % 3][0] = *reinterpret_cast<const float2*>(&Us[pos * 512 + urow]);
        uf[pos % 3][1] = *reinterpret_cast<const float2*>(&Us[pos * 512 + urow + 128]);
    };

    for (int k = 0; k < nunits; ++k) {
        const bool has1 = k + 1 < nunits, has2 = k + 2 < nunits;
        if (has2) { c2 = c1; advance(c2); }
        const bool np1 = has1 && new_patch(cc, c1);
        const int ub2 = ub == 0 ? 2 : ub - 1;       // buffer of U(k + 2) = the one U(k - 1) has left
        // the other patch buffer was last read by the transform of a unit < k: behind the barrier that opened this unit
        if (np1) issue_patch(c1, pb ^ 1);
        const float* Us = Ubuf + ub * kWinoUFloats;
        f32x2 V[16];
        if (!(kWinoDbg & 16)) {
            load_raw(pb, cc.s & 1, V);
#pragma unroll
            for (int hs = 0; hs < 16; ++hs) transform_half(V, hs);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) V[i] = f32x2{1.f, 1.f};
        }
        if (CLSLAM_WINO_TRACE >= 3) { asm volatile("" : "+v"(V[0]), "+v"(V[15])); stamp(); }
        float2 uf[3][2];
        uf_load(Us, uf, 0);
        uf_load(Us, uf, 1);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            if (pos + 2 < 16) uf_load(Us, uf, pos + 2);
            if (!(kWinoDbg & 4)) {
                const float2 a0 = uf[pos % 3][0], a1 = uf[pos % 3][1];
                const f32x2 bq = V[pos];
                acc[pos][0] = mfma_16x16x4(a0.x, bq[0], acc[pos][0]);
                acc[pos][1] = mfma_16x16x4(a1.x, bq[0], acc[pos][1]);
                acc[pos][0] = mfma_16x16x4(a0.y, bq[1], acc[pos][0]);
                acc[pos][1] = mfma_16x16x4(a1.y, bq[1], acc[pos][1]);
            }
            if (pos == kUAfter - 1 && has2) issue_u(c2, ub2);
            sched_fence();
        }
        if (CLSLAM_WINO_TRACE) stamp();
        bool drained = false;
        if (cc.s + 1 >= cc.hi) {
            drained = finish(cc.t, seg_lo(cc.t), cc.hi);
            if (has1) zero_acc();
            // (free here -- finish() has just drained -- and it leaves hipcc's wait-count pass with nothing pending on ANY path
            // out of finish(): otherwise it drains vmcnt(0), i.e. the DMA just issued, at the first LDS read of the next unit)
            vmem_drain_visible();
        }
        if (CLSLAM_WINO_TRACE) stamp();
        // every wave's DMA pieces of U(k+1) and of the patch of unit k+1 have landed behind this wait and the barrier; the four
        // pieces of U(k+2), the youngest group, stay in flight
        if (publish_armed) publish();
        else if (!drained) { if (has2) dma_wait_keep4(); else dma_wait_all(); }
        publish_armed = publish_pending; publish_pending = false;
        if (CLSLAM_WINO_TRACE >= 3) stamp();
        wg_barrier_keep_dma();
        if (CLSLAM_WINO_TRACE) stamp();
        pb ^= np1 ? 1 : 0;
        ub = ub == 2 ? 0 : ub + 1;
        cc = c1; c1 = c2;
    }
    if (publish_armed || publish_pending) publish();
}

// ---- U = G g G^T, written as the LDS image of the kernel's stages --------------------------------------------------------
__global__ void wino_weight_transform_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int tilesN, int NS) {
    const size_t total = (size_t)tilesN * NS * (kWinoUFloats / 4);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        // float4 `idx` of the LDS image: [block = (channel tile, stage)][pos 16][channel group 4][input-channel pair 4][8 float4], a
        // float4 = (channel n0: c0, c1)(channel n0 + 1: c0, c1)
        const int r = (int)(idx & 127);
        const int pos = (int)((idx >> 7) & 15);
        const size_t blk = idx >> 11;
        const int s = (int)(blk % NS), tn = (int)(blk / NS);
        const int cp = (r & 31) >> 3, n0 = tn * 64 + (r >> 5) * 16 + (r & 7) * 2;
        const int i = pos >> 2, j = pos & 3;
        float o[4];
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + (e >> 1), c = s * 8 + cp * 2 + (e & 1);
            o[e] = 0.f;
            if (n < Cout) {
                double g[3][3];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) g[a][b] = (double)w[((size_t)n * 9 + a * 3 + b) * Cin + c];
                // rows of G: (1,0,0), (1/2,1/2,1/2), (1/2,-1/2,1/2), (0,0,1)
                double rr[3];
                for (int b = 0; b < 3; ++b)
                    rr[b] = i == 0 ? g[0][b] : i == 1 ? 0.5 * (g[0][b] + g[1][b] + g[2][b]) : i == 2 ? 0.5 * (g[0][b] - g[1][b] + g[2][b]) : g[2][b];
                const double val = j == 0 ? rr[0] : j == 1 ? 0.5 * (rr[0] + rr[1] + rr[2]) : j == 2 ? 0.5 * (rr[0] - rr[1] + rr[2]) : rr[2];
                o[e] = (float)val;
            }
        }
        const float4 v = make_float4(o[0], o[1], o[2], o[3]);
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
           (d->act == CLSLAM_ACT_NONE || d->act == CLSLAM_ACT_RELU) &&
           d->out_h == d->in_h + 2 * d->pad - 2 && d->out_w == d->in_w + 2 * d->pad - 2;
}

// (tile, stage) units per persistent workgroup: what decides whether this kernel pays.  Its fixed costs per launch -- prologue, one
// hand-off per workgroup, ~7500 cycles per finish() -- are ~20-30 us; a stage costs ~3 us.  Measured on MI355X (tools/bench_conv.py,
// profiles/r05_wino_microbench.txt): >= 8 units per workgroup (the pose encoder's 2B images, every K >= 8 / 384x1280 workload)
// 93-110 TFLOP/s against 67-89 for the direct kernels; ~5 units (the depth encoder at B = 5) 55-63 against 67-80.
// Workgroups of a launch: one per CU the descriptor grants it (clslam_conv_desc.cu_limit; CLSLAM_WINO_GROUPS / CLSLAM_SK_GROUPS
// override it for experiments).
static int wino_groups(const clslam_conv_desc* d) {
    int g = sk_device_cus();
    if (d->cu_limit > 0) g = std::min(g, d->cu_limit);
    // experiment / test knobs, read per launch like conv_sk.hip's (tests cut the unit stream differently from case to case)
    if (const char* e = getenv("CLSLAM_SK_GROUPS")) g = std::max(1, atoi(e));
    if (const char* e = getenv("CLSLAM_WINO_GROUPS")) g = std::max(1, atoi(e));
    return std::min(g, kWinoMaxGroups);
}

int conv3x3_wino_units_per_group(const clslam_conv_desc* d) {
    int RB = 1, RH = 1, RW = 1;
    wino_pick_region(d->batch, d->out_h, d->out_w, RB, RH, RW);
    const long long tiles = (long long)cdiv(d->batch, RB) * cdiv((d->out_h + 1) / 2, RH) * cdiv((d->out_w + 1) / 2, RW) * cdiv(d->ch_out, 64);
    const long long units = tiles * (d->ch_a / 8);
    return (int)(units / std::max(1, wino_groups(d)));
}

// Called by clslam_conv2d for config 40.
int conv3x3_wino_dispatch(const clslam_conv_desc* d, hipStream_t stream) {
    if (!conv3x3_wino_supported(d)) {
        set_error("conv2d: the Winograd kernel needs a 3x3 stride-1 zero-padded conv of one source with weight_wino set");
        return CLSLAM_ERR_INVALID;
    }
    if ((long long)d->batch * d->out_h * d->out_w * d->ch_out >= (1ll << 31) || (long long)d->batch * d->in_h * d->in_w * d->ch_a >= (1ll << 31)) {
        set_error("conv2d: the Winograd kernel indexes its tensors with 32-bit element offsets");
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
    if ((long long)k.tiles * k.NS >= (1ll << 31)) { set_error("conv2d: too many (region, stage) units for the Winograd kernel"); return CLSLAM_ERR_INVALID; }
    k.units = k.tiles * k.NS;
    k.b_fastest = ((size_t)k.Cout * 16 * k.Cin > (size_t)k.B * k.Hi * k.Wi * k.Cin) ? 1 : 0;
    if (const char* e = getenv("CLSLAM_SK_B_FASTEST")) k.b_fastest = atoi(e);
    const int G = std::min(wino_groups(d), k.units);
    k.G = G;
    const size_t need = (size_t)kWinoSlabOffsetBytes + (size_t)G * kWinoSlabFloats * sizeof(float) + (CLSLAM_WINO_TRACE ? (size_t)G * 64 * 8 * (CLSLAM_WINO_TRACE >= 3 ? 8 : 1) : 0);
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
        hipExtLaunchKernelGGL(conv3x3_wino8_kernel, dim3(G), dim3(512), 0, stream, e0, e1, 0, k);
        return check_launch("conv3x3_wino");
    }
#endif
    hipLaunchKernelGGL(conv3x3_wino8_kernel, dim3(G), dim3(512), 0, stream, k);
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
