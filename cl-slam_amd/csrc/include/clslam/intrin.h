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

// 1/x to 1 ulp (v_rcp_f32) -- for the SSIM / projection quotients of the loss kernels, where an IEEE
// division costs ~10 instructions and the last ulp is far below the 1e-4 parity bar
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// value of lane (lane ^ mask)
__device__ __forceinline__ float wave_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }

}  // namespace clslam
