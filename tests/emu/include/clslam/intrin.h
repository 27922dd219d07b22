// EMULATED counterparts of cl-slam_amd/csrc/include/clslam/intrin.h (wave collectives and
// MFMA), following the gfx950 operand layouts documented in cdna_hip_programming.md section 3:
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r -> row 4*(l>>4)+r, col l&15
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950: two lanes' worth of work per issue slot)
inline f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
inline f32x2 pk_abs(f32x2 a) { return __builtin_elementwise_abs(a); }
inline f32x2 pk_max(f32x2 a, float b) { const f32x2 v = {b, b}; return __builtin_elementwise_max(a, v); }
inline f32x2 pk_min(f32x2 a, float b) { const f32x2 v = {b, b}; return __builtin_elementwise_min(a, v); }

void emu_yield_os();   // emu_runtime.cpp

namespace clslam {

constexpr int kWave = 64;
inline int lane_id() { return (int)(threadIdx.x & 63); }

inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    const int l = lane_id();
    const float* s = emu_wave_exchange2(a, b);
    const int j = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(s[k * 32 + i], s[64 + k * 32 + j], acc);
        c[r] = acc;
    }
    return c;
}

inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    const int l = lane_id();
    const float* s = emu_wave_exchange2(a, b);
    const int j = l & 15, g = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(s[k * 16 + i], s[64 + k * 16 + j], acc);
        c[r] = acc;
    }
    return c;
}

// butterfly all-reduce over the 64 lanes of a wave (same pairing order as the HIP version)
inline float wave_sum(float v) {
    const int l = lane_id();
    for (int off = 32; off >= 1; off >>= 1) {
        float* s = emu_wave_scratch() + emu_wave_phase() * 128;
        s[l] = v;
        emu_sync_wave();
        v = v + s[l ^ off];
    }
    return v;
}

inline double wave_sum_f64(double v) {
    const int l = lane_id();
    for (int off = 32; off >= 1; off >>= 1) {
        double* s = reinterpret_cast<double*>(emu_wave_scratch() + emu_wave_phase() * 128);
        s[l] = v;
        emu_sync_wave();
        v = v + s[l ^ off];
    }
    return v;
}

inline float fast_rcp(float x) { return 1.f / x; }

// blocks run on several host threads: real atomics / fences
inline void coherent_store(float* p, float v) { __atomic_store(p, &v, __ATOMIC_RELAXED); }
inline float coherent_load(const float* p) { float v; __atomic_load(p, &v, __ATOMIC_RELAXED); return v; }
inline void coherent_store_u32(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline unsigned coherent_inc(unsigned* p) { return __atomic_fetch_add(p, 1u, __ATOMIC_SEQ_CST); }
inline void stores_complete() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void consume_now(float&) {}

inline unsigned coherent_load_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline void spin_pause() { ::emu_yield_os(); }   // the producer block runs on another host thread
// LDS-DMA emulated as an immediate copy: lane l's 16 bytes land at lds_wave_base + l * 16
inline void lds_dma16(const float* gsrc, float* lds_wave_base) { memcpy(lds_wave_base + lane_id() * 4, gsrc, 16); }
inline unsigned lds_addr(const float*) { return 0u; }
inline void launder_uniform(unsigned&) {}
inline f32x2 lds_read_f32x2(const float* lds_array, unsigned, unsigned byte_offset) { f32x2 v; memcpy(&v, (const char*)lds_array + byte_offset, 8); return v; }
inline void lds_dma16_at(const float* gsrc, float* lds_array, unsigned, unsigned float_offset) { memcpy(lds_array + float_offset + lane_id() * 4, gsrc, 16); }
inline void lds_dma16_x4(const float* gsrc, float* lds_array, unsigned, unsigned float_offset) { for (int i = 0; i < 4; ++i) memcpy(lds_array + float_offset + i * 256 + lane_id() * 4, gsrc + i * 256, 16); }
inline void dma_wait_all() {}
inline void dma_wait_keep8() {}
inline void dma_wait_keep4() {}
inline void sched_fence() {}
inline int launder(int v) { return v; }
inline void pin4(float4&, float4&, float4&, float4&) {}
inline void pin2(float4&, float4&) {}
inline void pin1(float4&) {}
inline void uncounted_flag_store(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline void uncounted_store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
inline void mfma_results_settle() {}
inline void vmem_drain_visible() {}
inline void coherent_store4(float* p, f32x4 v) { memcpy(p, &v, 16); __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void coherent_load4x4(const float* p, f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    memcpy(&a, p, 16); memcpy(&b, p + 256, 16); memcpy(&c, p + 512, 16); memcpy(&d, p + 768, 16);
}
inline void coherent_load4x8(const float* p, f32x4 (&v)[8]) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    for (int i = 0; i < 8; ++i) memcpy(&v[i], p + 256 * i, 16);
}
inline void coherent_load4x2(const float* p, f32x4& a, f32x4& b) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    memcpy(&a, p, 16); memcpy(&b, p + 256, 16);
}
inline void coherent_load4x4_x4(const float* p0, const float* p1, const float* p2, const float* p3,
                                f32x4 (&a)[4], f32x4 (&b)[4], f32x4 (&c)[4], f32x4 (&d)[4]) {
    coherent_load4x4(p0, a[0], a[1], a[2], a[3]); coherent_load4x4(p1, b[0], b[1], b[2], b[3]);
    coherent_load4x4(p2, c[0], c[1], c[2], c[3]); coherent_load4x4(p3, d[0], d[1], d[2], d[3]);
}
inline void coherent_load4x2_x4(const float* p0, const float* p1, const float* p2, const float* p3,
                                f32x4 (&a)[2], f32x4 (&b)[2], f32x4 (&c)[2], f32x4 (&d)[2]) {
    coherent_load4x2(p0, a[0], a[1]); coherent_load4x2(p1, b[0], b[1]);
    coherent_load4x2(p2, c[0], c[1]); coherent_load4x2(p3, d[0], d[1]);
}
inline void wg_barrier_keep_dma() { emu_sync_block(); }

inline float wave_shfl_xor(float v, int mask) {
    const int l = lane_id();
    float* s = emu_wave_scratch() + emu_wave_phase() * 128;
    s[l] = v;
    emu_sync_wave();
    return s[l ^ mask];
}

}  // namespace clslam
