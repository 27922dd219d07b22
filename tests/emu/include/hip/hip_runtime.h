// CPU emulation of the small subset of the HIP programming model the clslam kernels use.
// TEST INFRASTRUCTURE ONLY (tests/emu): lets the kernel *sources* under cl-slam_amd/csrc be
// compiled for the host and executed block-by-block with one fiber per GPU thread, so that
// indexing / math / host sequencing can be validated in a container without a GPU.  It is
// never loaded by the product (cl-slam_amd/clslam_hip/_lib.py only loads libclslam_hip.so).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
enum { hipMemcpyDeviceToDevice = 3 };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }

// Barrier across the block's fibers / across one 64-lane wave (emu_runtime.cpp).
void emu_sync_block();
void emu_sync_wave();
float* emu_wave_exchange2(float a, float b);   // both MFMA operands of this lane out, the wave's exchange buffer back
float* emu_wave_scratch();  // 2 x 64 x 2 floats of per-wave exchange space, double-buffered
int emu_wave_phase();       // flips at each wave-collective
inline void __syncthreads() { emu_sync_block(); }

void emu_launch(std::function<void()> body, dim3 grid, dim3 block);

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/,
                               hipStream_t /*stream*/, Args... args) {
    emu_launch([=]() { kernel(static_cast<KArgs>(args)...); }, grid, block);
}

using std::min;
using std::max;
inline float __expf(float x) { return expf(x); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline float __fdividef(float a, float b) { return a / b; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline float atomicAdd(float* p, float v) {
    float old = 0.f;
    // fp atomic add via CAS on the bit pattern
    uint32_t* ip = reinterpret_cast<uint32_t*>(p);
    uint32_t o, n;
    do {
        o = __atomic_load_n(ip, __ATOMIC_RELAXED);
        float f; memcpy(&f, &o, 4); old = f; f += v; memcpy(&n, &f, 4);
    } while (!__atomic_compare_exchange_n(ip, &o, n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return old;
}
