// Calibration micro-benchmark (measurement only): s_memtime ticks per v_mfma_f32_32x32x2_f32, independent vs chained accumulators,
// with and without VALU fillers between the MFMAs, one wave per SIMD on every CU.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock tools/micro/mfma_clock.hip && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    float f0 = x, f1 = y, f2 = x + y, f3 = x - y, f4 = 1.f, f5 = 2.f, f6 = 3.f, f7 = 4.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {          // four independent accumulators
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        } else if constexpr (MODE == 1) {   // one chain
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        } else if constexpr (MODE >= 100) {  // TWO alternating accumulators (chains two apart) with (MODE-100) VALU adds between the MFMAs
            for (int t = 0; t < 4; ++t) {
                if (t & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < (MODE - 100) / 8; ++q) {
                    f0 += f4; f1 -= f5; f2 += f6; f3 -= f7; f4 += f0; f5 -= f1; f6 += f2; f7 -= f3;
                    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
                }
            }
        } else {                            // chain with MODE-2 VALU adds between the MFMAs
            for (int t = 0; t < 4; ++t) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < (MODE - 2) / 8; ++q) {
                    f0 += f4; f1 -= f5; f2 += f6; f3 -= f7; f4 += f0; f5 -= f1; f6 += f2; f7 -= f3;
                    asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.f) sink[0] = s;
}

template <int MODE>
void run(const char* name, unsigned long long* d, float* sink) {
    const int iters = 4096, G = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<G, 256>>>(d, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<G, 256>>>(d, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(G);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= G;
    const double n = iters * 4.0;
    printf("%-34s %8.1f us  ticks/MFMA %6.2f  ticks/us %7.1f  => %5.1f TFLOP/s\n", name, ms * 1e3, avg / n, avg / (ms * 1e3),
           n * 4 * 256 * 2.0 * 32 * 32 * 2 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 4096 * 8); hipMalloc(&sink, 4);
    run<0>("4 independent accumulators", d, sink);
    run<1>("one chain", d, sink);
    run<10>("chain + 8 VALU per gap", d, sink);
    run<18>("chain + 16 VALU per gap", d, sink);
    run<34>("chain + 32 VALU per gap", d, sink);
    run<100>("2 alternating accumulators", d, sink);
    run<108>("2 alternating + 8 VALU per gap", d, sink);
    run<116>("2 alternating + 16 VALU per gap", d, sink);
    run<124>("2 alternating + 24 VALU per gap", d, sink);
    return 0;
}
