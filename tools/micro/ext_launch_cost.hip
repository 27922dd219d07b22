// Host-side price of hipExtLaunchKernelGGL with a stop event (ROCm 7.2, gfx950): N launches of a ~10 us kernel on one stream,
// host enqueue time per launch (before any synchronisation) and total time per launch.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ext_launch_cost tools/micro/ext_launch_cost.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[threadIdx.x] = v;
}

int main() {
    const int N = 200, iters = 600;
    float* a;
    (void)hipMalloc(&a, 4096); (void)hipMemset(a, 0, 4096);
    hipStream_t s1;
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    std::vector<hipEvent_t> nt(N), tm(N), tm2(N);
    for (int i = 0; i < N; ++i) {
        (void)hipEventCreateWithFlags(&nt[i], hipEventDisableTiming);
        (void)hipEventCreate(&tm[i]);
        (void)hipEventCreate(&tm2[i]);
    }
    const char* names[] = {"hipLaunchKernelGGL", "ext, no events", "ext, stop event (timing disabled)", "ext, stop event (default flags)",
                           "ext, start + stop events", "hipLaunchKernelGGL + hipEventRecord (timing disabled)",
                           "ext, stop event on every 4th launch"};
    for (int pass = 0; pass < 2; ++pass)
    for (int mode = 0; mode < 7; ++mode) {
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, a, iters); break;
                case 1: hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, nullptr, nullptr, 0, a, iters); break;
                case 2: hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, nullptr, nt[i], 0, a, iters); break;
                case 3: hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, nullptr, tm[i], 0, a, iters); break;
                case 4: hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, tm2[i], tm[i], 0, a, iters); break;
                case 5: hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, a, iters); (void)hipEventRecord(nt[i], s1); break;
                case 6: if (i % 4 == 0) hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, nullptr, nt[i], 0, a, iters);
                        else hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s1, a, iters);
                        break;
            }
        }
        auto t1 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(s1);
        auto t2 = std::chrono::steady_clock::now();
        if (pass)
            printf("%-56s host %.2f us per launch, total %.2f us per launch\n", names[mode],
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
                   std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    return 0;
}
