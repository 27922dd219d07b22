// Calibration (measurement only): LDS-DMA (global_load_lds_dwordx4) throughput per CU with every CU streaming, by source locality.
//   MODE 0: all workgroups read the SAME 256 KiB (L2 hits: the Winograd filter image of a 64-channel layer)
//   MODE 1: every workgroup streams its own region of a 256 MiB buffer (Infinity Cache / HBM)
// Each iteration moves `kb` KiB per workgroup (one 1 KiB piece per wave-instruction), then waits and barriers.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldr tools/micro/lds_dma_rate.hip && /tmp/ldr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma(const float* g, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
template <int NT>
__global__ __launch_bounds__(NT) void k(const float* src, unsigned long long* out, int iters, int kb, size_t wg_stride, size_t span) {
    __shared__ __attribute__((aligned(1024))) float lds[32768];      // 128 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = NT / 64;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
    const float* base = src + (size_t)blockIdx.x * wg_stride;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    size_t off = 0;
    for (int i = 0; i < iters; ++i) {
        for (int pc = wave; pc < kb; pc += NW) dma(base + (off + (size_t)pc * 256) % span + lane * 4, lds0 + (pc % 128) * 1024);
        off += (size_t)kb * 256;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (lds[threadIdx.x] == 12345.f) out[0] = 0;
}
template <int NT>
void run(const char* name, const float* src, unsigned long long* d, int kb, size_t wg_stride, size_t span) {
    const int iters = 200, G = 256;
    k<NT><<<G, NT>>>(src, d, iters, kb, wg_stride, span);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NT><<<G, NT>>>(src, d, iters, kb, wg_stride, span);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(G);
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= G;
    printf("%-46s %3d KiB/iter  %7.0f cycles/iter  %6.1f B/clk/CU  %6.2f TB/s chip\n", name, kb, avg / iters, kb * 1024.0 / (avg / iters),
           (double)G * kb * 1024.0 * iters / (ms * 1e-3) / 1e12);
}
int main() {
    float* src; unsigned long long* d;
    const size_t total = (size_t)256 << 20;
    hipMalloc(&src, total); hipMemset(src, 0, total); hipMalloc(&d, 4096 * 8);
    const size_t f = total / 4;
    for (int kb : {16, 43, 86}) {
        run<256>("4 waves, shared 256 KiB (L2)", src, d, kb, 0, 65536);
        run<512>("8 waves, shared 256 KiB (L2)", src, d, kb, 0, 65536);
        run<512>("8 waves, shared 4 MiB", src, d, kb, 0, 1 << 20);
        run<512>("8 waves, own 1 MiB region each (256 MiB total)", src, d, kb, f / 256, f / 256);
    }
    return 0;
}
