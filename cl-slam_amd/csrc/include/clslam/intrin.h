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

// butterfly all-reduce over the 64 lanes (deterministic pairing order)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// value of lane (lane ^ mask)
__device__ __forceinline__ float wave_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }

}  // namespace clslam
