// What a cross-stream hand-off costs the PRODUCER stream on gfx950 / ROCm 7.2 (tools/measure_round.sh -> profiles/*_micro_calibration.txt).
// A chain of N dependent ~10 us kernels on stream s1; after every kernel a second stream s2 is released to run a tiny kernel:
//   mode 0: no hand-off at all (the chain alone)
//   mode 1: hipEventRecord(ev, s1) + hipStreamWaitEvent(s2, ev)            -- what torch.cuda.Event().record(stream) does
//   mode 2: the event is the kernel's OWN completion signal: hipExtLaunchKernelGGL(..., stopEvent = ev) + hipStreamWaitEvent(s2, ev)
// Prints the chain's time per link.  hipcc --offload-arch=gfx950 -O3 -o /tmp/event_gap tools/micro/event_gap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}
__global__ void tiny_kernel(float* p) { p[threadIdx.x] += 1.f; }
// ordering check of mode 2: the producer publishes its link number at its very END, the consumer copies what it sees
__global__ void produce_kernel(float* p, int iters, int* flag, int link) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *flag = link;
}
__global__ void consume_kernel(const int* flag, int* seen, int link) { seen[link] = *flag; }

int main(int argc, char** argv) {
    const int N = 200;
    float *a, *b;
    hipMalloc(&a, 4096); hipMalloc(&b, 4096);
    hipMemset(a, 0, 4096); hipMemset(b, 0, 4096);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int iters = argc > 1 ? atoi(argv[1]) : 600;   // 600: ~10 us
    for (int mode : {0, 1, 2, 0, 1, 2}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 2) {
                    hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, nullptr, ev[i], 0, a, iters);
                } else {
                    hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, a, iters);
                    if (mode == 1) hipEventRecord(ev[i], s1);
                }
                if (mode != 0) {
                    hipStreamWaitEvent(s2, ev[i], 0);
                    hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s2, b);
                }
            }
            auto th = std::chrono::steady_clock::now();
            hipStreamSynchronize(s1);
            auto t1 = std::chrono::steady_clock::now();
            hipDeviceSynchronize();
            if (rep == 2)
                printf("mode %d: %.2f us per link of the s1 chain (%d links; the host needed %.2f us per link to enqueue)\n", mode,
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / N, N,
                       std::chrono::duration<double, std::micro>(th - t0).count() / N);
        }
    }
    // does a wait on a stop event really order the consumer behind the producer's END?
    int *flag, *seen;
    hipMalloc(&flag, 4); hipMalloc(&seen, 4 * N);
    hipMemset(flag, 0xff, 4); hipMemset(seen, 0xff, 4 * N);
    hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) {
        hipExtLaunchKernelGGL(produce_kernel, dim3(64), dim3(256), 0, s1, nullptr, ev[i], 0, a, iters, flag, i);
        hipStreamWaitEvent(s2, ev[i], 0);
        hipLaunchKernelGGL(consume_kernel, dim3(1), dim3(1), 0, s2, flag, seen, i);
    }
    hipDeviceSynchronize();
    std::vector<int> h(N);
    hipMemcpy(h.data(), seen, 4 * N, hipMemcpyDeviceToHost);
    int early = 0;
    for (int i = 0; i < N; ++i) early += h[i] < i;
    printf("stop-event ordering: %d of %d consumers ran before their producer had finished (%s)\n", early, N, early ? "BROKEN" : "ok");
    printf("last hip error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
