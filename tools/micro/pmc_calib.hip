// Calibration (measurement only) of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of conv_wino.hip
// (MI355X_MICROARCH.md: "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every kernel moves exactly 64 MiB of USEFUL bytes out of / into a 256 MiB buffer (past the L2s; the writes of 64 MiB
// spread over 256 MiB where the pattern is strided):
//   wr_coalesced   16 B per lane, a wave writes 1 KiB contiguous
//   wr_seg64       16 B per lane, groups of four lanes write a 64-B segment, segments 512 B apart (the Winograd epilogue: four
//                  channels x four lane groups of a pixel, tiles two pixels of 64 channels apart)
//   wr_through     as wr_coalesced with sc0 sc1 (the partial slabs of the stream-K hand-off)
//   rd_coalesced   16 B per lane, a wave reads 1 KiB contiguous (LDS-DMA: global_load_lds_dwordx4)
//   rd_seg64       LDS-DMA, groups of four lanes read a 64-B segment, segments 256 B apart (the 16-channel chunk of a 64-channel
//                  pixel: the patch rows of layer1)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib tools/micro/pmc_calib.hip
//   rocprofv3 --pmc WRITE_SIZE -d /tmp/pc_w -o run -- /tmp/pmc_calib ;  rocprofv3 --pmc FETCH_SIZE -d /tmp/pc_f -o run -- /tmp/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr size_t kUseful = (size_t)64 << 20, kLanes = kUseful / 16;
__global__ void wr_coalesced(float* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 v = {1.f, 2.f, 3.f, (float)i};
    *reinterpret_cast<f32x4*>(dst + i * 4) = v;
}
__global__ void wr_seg64(float* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 v = {1.f, 2.f, 3.f, (float)i};
    *reinterpret_cast<f32x4*>(dst + (i >> 2) * 128 + (i & 3) * 4) = v;          // 64-B segment every 512 B
}
__global__ void wr_through(float* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 v = {1.f, 2.f, 3.f, (float)i};
    float* p = dst + i * 4;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void dma(const float* g, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
template <bool SEG>
__global__ __launch_bounds__(256) void rd_kernel(const float* src, float* sink) {
    __shared__ __attribute__((aligned(1024))) float lds[1024];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* g = SEG ? src + (i >> 2) * 64 + (i & 3) * 4 : src + i * 4;      // SEG: 64-B segment every 256 B
    dma(g, lds0 + wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 12345.f) sink[0] = 1.f;
}
int main() {
    float *buf, *sink;
    hipMalloc(&buf, (size_t)512 << 20); hipMemset(buf, 0, (size_t)512 << 20); hipMalloc(&sink, 64);
    const int nb = (int)(kLanes / 256);
    for (int rep = 0; rep < 3; ++rep) {
        wr_coalesced<<<nb, 256>>>(buf);
        wr_seg64<<<nb, 256>>>(buf);
        wr_through<<<nb, 256>>>(buf);
        rd_kernel<false><<<nb, 256>>>(buf, sink);
        rd_kernel<true><<<nb, 256>>>(buf, sink);
        hipDeviceSynchronize();
    }
    printf("every kernel: %zu useful bytes\n", kUseful);
    return 0;
}
