// Calibration (measurement only): does v_mfma_f32_32x32x2_f32 share execution resources with plain fp32 VALU work of ANOTHER wave on
// the same SIMD?  512-thread workgroups (two waves per SIMD): waves 0-3 run an MFMA chain, waves 4-7 run MODE: 0 nothing (exit),
// 1 a v_fma_f32 stream, 2 a ds_read_b128 stream, 3 an s_nop stream.  Reports the MFMA waves' cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvs tools/micro/mfma_valu_share.hip && /tmp/mvs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.f;
    __syncthreads();
    if (wave < 4) {
        f32x16 a0 = {}, a1 = {};
        float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
        float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
        if (s == 12345.f) sink[0] = s;
    } else if (MODE == 1) {
        float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, f6 = 6.f, f7 = 7.f;
        for (int i = 0; i < iters * 8; ++i) {      // independent v_fma streams, ~ the MFMA waves' duration
            f0 = f0 * 1.0001f + 1.f; f1 = f1 * 1.0001f + 1.f; f2 = f2 * 1.0001f + 1.f; f3 = f3 * 1.0001f + 1.f;
            f4 = f4 * 1.0001f + 1.f; f5 = f5 * 1.0001f + 1.f; f6 = f6 * 1.0001f + 1.f; f7 = f7 * 1.0001f + 1.f;
        }
        if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 12345.f) sink[1] = f0;
    } else if (MODE == 2) {
        float4 acc = {0, 0, 0, 0};
        const float4* l4 = reinterpret_cast<const float4*>(lds);
        for (int i = 0; i < iters * 4; ++i) {
            const float4 v0 = l4[(threadIdx.x + i) & 1023], v1 = l4[(threadIdx.x + i + 64) & 1023];
            acc.x += v0.x + v1.x; acc.y += v0.y + v1.y;
        }
        if (acc.x + acc.y == 12345.f) sink[2] = acc.x;
    } else if (MODE == 3) {
        for (int i = 0; i < iters * 8; ++i) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
    }
}
template <int MODE>
void run(const char* name, unsigned long long* d, float* sink) {
    const int iters = 4096, G = 256;
    k<MODE><<<G, 512>>>(d, sink, iters);
    hipDeviceSynchronize();
    k<MODE><<<G, 512>>>(d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(G * 4);
    hipMemcpy(h.data(), d, G * 4 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= (G * 4);
    printf("%-44s cycles per MFMA (MFMA waves) %7.2f\n", name, avg / (iters * 4.0));
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 4096 * 8); hipMalloc(&sink, 16);
    run<0>("partner wave: idle", d, sink);
    run<1>("partner wave: v_fma_f32 stream", d, sink);
    run<2>("partner wave: ds_read_b128 + few VALU", d, sink);
    run<3>("partner wave: s_nop stream", d, sink);
    return 0;
}
