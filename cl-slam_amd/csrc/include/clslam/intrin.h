// gfx950 (CDNA4) wave-level primitives used by the clslam kernels.
//   fp32-input MFMA: v_mfma_f32_32x32x2_f32 (64 cyc/SIMD) and v_mfma_f32_16x16x4_f32 (32 cyc/SIMD),
//   exact fp32 (bitwise an fmaf chain) at the 157 TFLOP/s fp32 matrix rate.  bf16/fp8 MFMA is
//   not used: the parity bar of the path is 1e-4 relative on depth and pose.
// Operand layout (lane l of the 64-wide wavefront):
//   32x32x2 : a = A[i=l&31][k=l>>5], b = B[k=l>>5][j=l&31]; D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
//   16x16x4 : a = A[i=l&15][k=l>>4], b = B[k=l>>4][j=l&15]; D reg r -> row 4*(l>>4)+r,              col l&15
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950: two lanes' worth of work per issue slot)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_abs(f32x2 a) { return __builtin_elementwise_abs(a); }
__device__ __forceinline__ f32x2 pk_max(f32x2 a, float b) { const f32x2 v = {b, b}; return __builtin_elementwise_max(a, v); }
__device__ __forceinline__ f32x2 pk_min(f32x2 a, float b) { const f32x2 v = {b, b}; return __builtin_elementwise_min(a, v); }

namespace clslam {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// DPP lane permutation inside a 16-lane row (quad_perm / row_mirror / row_half_mirror controls)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// Sum over the 64 lanes, returned in every lane; call from wave-uniform control flow only.  Four DPP
// adds (VALU, no LDS traffic -- __shfl_xor is a ds_bpermute per step) leave every lane with the sum of
// its 16-lane row, the four row sums are read with v_readlane and added on the scalar side.  Fixed
// pairing order: deterministic.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm:[1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm:[2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}

// The same in double: the 24 pose-gradient sums of the loss backward (a few million terms of both signs per sample, and
// K^T dP cancels |u| ~ 600 against them afterwards) are reduced in double from the wave level on -- 24 numbers per block,
// free.  Each 64-bit value travels as two DPP moves.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Cross-workgroup hand-off inside one kernel (split-K gather).  MI355X has one L2 per XCD and they are
// not coherent with each other for ordinary accesses; a device-scope __threadfence() makes them so by
// writing back / invalidating the whole L2 (measured: ~300 us per conv when 1280 workgroups do it).
// Agent-scope relaxed atomics instead carry the scope on the instruction itself (sc1): the store is
// written through to the coherence point, the load bypasses non-coherent lines -- no cache-wide
// maintenance.  stores_complete() waits until this lane's stores have been acknowledged.
__device__ __forceinline__ void coherent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float coherent_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned coherent_inc(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_complete() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// The value is needed HERE as far as the compiler is concerned: the wait for its load is placed at this point instead of inside
// every conditional block that uses it later (where, stores counting in vmcnt on gfx9, `s_waitcnt vmcnt(n)` throttles the
// stores of an epilogue to n in flight).
__device__ __forceinline__ void consume_now(float& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ unsigned coherent_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(8); }

// LDS-DMA (gfx950 global_load_lds_dwordx4): every lane fetches 16 bytes from ITS OWN global address and the wave's
// 64 x 16 B land at lds_wave_base + lane * 16 -- contiguous, no VGPR staging, no ds_write.  The transfer is
// asynchronous: it counts in vmcnt and nothing orders an LDS read behind it except the issuing wave's
// dma_wait_all() followed by a workgroup barrier for the other waves' reads (MI355X_MICROARCH.md, LDS-DMA).
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_wave_base) {
    // Inline asm on purpose: through __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) treats every DMA as a write
    // that may alias the next DMA and every ds_read of the same array, and drains vmcnt(0) between them -- the
    // transfers of a stage ran one after the other and the double buffering never overlapped.  M0 (destination
    // base) is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md, asm recipes).
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// The same with the destination as a wave-uniform LDS BYTE ADDRESS (lds_addr() of the array once + constant offsets: SALU only --
// converting a generic pointer per transfer costs a null check, two readfirstlanes and a 64-bit add each time)
__device__ __forceinline__ unsigned lds_addr(const float* lds_ptr) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const float*)lds_ptr);
}
// a wave-uniform value made opaque in a scalar register
__device__ __forceinline__ void launder_uniform(unsigned& v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); }
// 8-byte LDS read at lds_addr(array) + a BYTE offset the kernel computed itself (one v_xad_u32 for swizzle + buffer base; through
// a generic float* index hipcc forms the address with a shift-add per read)
__device__ __forceinline__ f32x2 lds_read_f32x2(const float*, unsigned lds_base_addr, unsigned byte_offset) {
    return *(const __attribute__((address_space(3))) f32x2*)(size_t)(lds_base_addr + byte_offset);
}
__device__ __forceinline__ void lds_dma16_at(const float* gsrc, float*, unsigned lds_base_addr, unsigned float_offset) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base_addr + float_offset * 4u);     // (uniform; not always provably so)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// Four consecutive 1 KiB pieces of a CONTIGUOUS source: the instruction offset advances the global address AND the LDS address
// (LDS address = M0 + offset + 16 * lane), so one M0 write and one address pair serve four transfers.
__device__ __forceinline__ void lds_dma16_x4(const float* gsrc, float*, unsigned lds_base_addr, unsigned float_offset) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base_addr + float_offset * 4u);     // (uniform; not always provably so)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// An MFMA result read by an INLINE-ASM memory instruction: hipcc's hazard recognizer pads its own instructions with the
// wait states gfx950 needs between an XDL write and a VMEM read of the same VGPR (up to 19), but does not look inside asm
// statements -- measured: the last lanes of the last MFMA's first result register reached a store stale.  Call once
// between the MFMAs and the first coherent_store4 of their accumulators.
__device__ __forceinline__ void mfma_results_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
// 16-byte agent-scope (write-through / L1-bypassing) accesses for cross-workgroup hand-offs inside one launch
__device__ __forceinline__ void coherent_store4(float* p, f32x4 v) {
#ifdef CLSLAM_SCALAR_HANDOFF
    for (int r = 0; r < 4; ++r) __hip_atomic_store(p + r, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    // s_nop 1: a store of more than 8 bytes reads its data VGPRs over the cycles after issue; the next instruction may
    // not overwrite them for two wait states.  hipcc pads its own stores, not asm ones -- with the accumulators in AGPRs
    // it re-fills the same four VGPRs (v_accvgpr_read) right behind each store and the last lanes stored the NEXT tile's
    // first register (found on hardware: 4-wave kernels only, the 8-wave ones store straight from the accumulator VGPRs).
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
#endif
}
// Plain 16-byte store that hipcc does not count: its wait-count pass would otherwise treat the store as pending at the
// next loop header and drain vmcnt(0) there -- together with the LDS-DMA in flight.  The kernel's own dma_wait_all()
// one unit later covers it.
__device__ __forceinline__ void uncounted_store4(float* p, float4 v) {
    f32x4 q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(q) : "memory");
}
// agent-scope flag store, equally uncounted (a 4-byte store has no data-read hazard)
__device__ __forceinline__ void uncounted_flag_store(unsigned* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
// four slab rows 1 KiB apart, one round trip
__device__ __forceinline__ void coherent_load4x4(const float* p, f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
#ifdef CLSLAM_SCALAR_HANDOFF
    for (int r = 0; r < 4; ++r) {
        a[r] = __hip_atomic_load(p + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b[r] = __hip_atomic_load(p + 256 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c[r] = __hip_atomic_load(p + 512 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d[r] = __hip_atomic_load(p + 768 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
#endif
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void coherent_load4x2(const float* p, f32x4& a, f32x4& b) {
#ifdef CLSLAM_SCALAR_HANDOFF
    for (int r = 0; r < 4; ++r) {
        a[r] = __hip_atomic_load(p + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b[r] = __hip_atomic_load(p + 256 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
#endif
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
// eight slab rows 1 KiB apart, one round trip
__device__ __forceinline__ void coherent_load4x8(const float* p, f32x4 (&v)[8]) {
#ifdef CLSLAM_SCALAR_HANDOFF
    coherent_load4x4(p, v[0], v[1], v[2], v[3]); coherent_load4x4(p + 1024, v[4], v[5], v[6], v[7]);
    return;
#endif
    const float* q = p + 1024;
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %8, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %8, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:3072 sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off sc0 sc1\n\tglobal_load_dwordx4 %5, %9, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:3072 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p), "v"(q) : "memory");
}
// The same rows of FOUR slabs in one round trip (the owner of a tile gathers its contributors four at a time).
__device__ __forceinline__ void coherent_load4x4_x4(const float* p0, const float* p1, const float* p2, const float* p3,
                                                    f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], f32x4 (&d)[4]) {
#ifdef CLSLAM_SCALAR_HANDOFF
    coherent_load4x4(p0, a[0], a[1], a[2], a[3]); coherent_load4x4(p1, b[0], b[1], b[2], b[3]);
    coherent_load4x4(p2, c[0], c[1], c[2], c[3]); coherent_load4x4(p3, d[0], d[1], d[2], d[3]);
    return;
#endif
    asm volatile("global_load_dwordx4 %0, %16, off sc0 sc1\n\tglobal_load_dwordx4 %1, %16, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %16, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:3072 sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %17, off sc0 sc1\n\tglobal_load_dwordx4 %5, %17, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %17, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %7, %17, off offset:3072 sc0 sc1\n\t"
                 "global_load_dwordx4 %8, %18, off sc0 sc1\n\tglobal_load_dwordx4 %9, %18, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %10, %18, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %11, %18, off offset:3072 sc0 sc1\n\t"
                 "global_load_dwordx4 %12, %19, off sc0 sc1\n\tglobal_load_dwordx4 %13, %19, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %14, %19, off offset:2048 sc0 sc1\n\tglobal_load_dwordx4 %15, %19, off offset:3072 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]),
                   "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ void coherent_load4x2_x4(const float* p0, const float* p1, const float* p2, const float* p3,
                                                    f32x4 (&a)[2], f32x4 (&b)[2], f32x4 (&c)[2], f32x4 (&d)[2]) {
#ifdef CLSLAM_SCALAR_HANDOFF
    coherent_load4x2(p0, a[0], a[1]); coherent_load4x2(p1, b[0], b[1]);
    coherent_load4x2(p2, c[0], c[1]); coherent_load4x2(p3, d[0], d[1]);
    return;
#endif
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %8, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %9, off sc0 sc1\n\tglobal_load_dwordx4 %3, %9, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %10, off sc0 sc1\n\tglobal_load_dwordx4 %5, %10, off offset:1024 sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %11, off sc0 sc1\n\tglobal_load_dwordx4 %7, %11, off offset:1024 sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(c[0]), "=&v"(c[1]), "=&v"(d[0]), "=&v"(d[1])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
// value made opaque at this point: address arithmetic that depends on it is neither hoisted out of the enclosing loop (where it
// would sit in -- or be spilled from -- dozens of VGPRs) nor merged with other uses
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }
// four float4 pinned at this point of the program order (between the volatile MFMA / DMA statements): hipcc otherwise SINKS work
// whose results are only needed later -- the interleaved input transform of conv_wino.hip ended up behind the last MFMA
__device__ __forceinline__ void pin4(float4& a, float4& b, float4& c, float4& d) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w), "+v"(c.x), "+v"(c.y),
                 "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
}
__device__ __forceinline__ void pin2(float4& a, float4& b) {
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
}
__device__ __forceinline__ void pin1(float4& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }
// nothing is scheduled across this point (pins the fragment reads of the next tap ahead of the current tap's MFMAs)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// vmcnt(0) THROUGH THE BUILTIN: unlike the asm form hipcc's wait-count pass sees it and knows that none of ITS
// loads/stores is pending afterwards (otherwise it drains vmcnt -- and with it the LDS-DMA in flight -- at the loop
// header before re-using a register that a store of the previous iteration read)
__device__ __forceinline__ void vmem_drain_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// all but the eight youngest vector-memory operations of this wave (loads retire in issue order: MI355X_MICROARCH.md)
__device__ __forceinline__ void dma_wait_keep4() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
__device__ __forceinline__ void dma_wait_keep8() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
// workgroup barrier WITHOUT the vmcnt(0) hipcc attaches to __syncthreads() while LDS-DMA is in flight: LDS
// reads/writes of this wave are drained (lgkmcnt), the DMA of the NEXT stage stays in flight across it
__device__ __forceinline__ void wg_barrier_keep_dma() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// 1/x to 1 ulp (v_rcp_f32) -- for the SSIM / projection quotients of the loss kernels, where an IEEE
// division costs ~10 instructions and the last ulp is far below the 1e-4 parity bar
#ifdef CLSLAM_EXACT_DIV      // diagnostic builds only (tools/diag_pose.py): IEEE division in place of v_rcp_f32
__device__ __forceinline__ float fast_rcp(float x) { return 1.f / x; }
#else
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

// value of lane (lane ^ mask)
__device__ __forceinline__ float wave_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }

}  // namespace clslam
