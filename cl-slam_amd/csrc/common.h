// Shared declarations for the clslam HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#if CLSLAM_DEVICE_BUILD
#include <hip/hip_ext.h>
#endif
#include <clslam/intrin.h>

#include <algorithm>

#include "../../include/clslam_hip.h"

namespace clslam {

// ---- error reporting across the C ABI (no exceptions cross it) -------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define CLSLAM_REQUIRE(cond, ...)                     \
    do {                                              \
        if (!(cond)) {                                \
            ::clslam::set_error(__VA_ARGS__);         \
            return CLSLAM_ERR_INVALID;                \
        }                                             \
    } while (0)

// ---- small device helpers -------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) {  // ReflectionPad2d semantics, |overhang| < n
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == CLSLAM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == CLSLAM_ACT_ELU) return v > 0.f ? v : expm1f(v);
    if (act == CLSLAM_ACT_HSWISH) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
    if (act == CLSLAM_ACT_HSIGMOID) return fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
    return v;
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_output(float y, int act) {
    if (act == CLSLAM_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == CLSLAM_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// XCD-aware block remap (MI355X: 8 XCDs, block b is dispatched to XCD b%8, each with a private
// 4 MiB L2).  Consecutive LOGICAL ids land on the same XCD so neighbouring tiles (same weight
// slice, adjacent pixels) share that L2.  Bijective for any nblk.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Measurement hook (clslam_conv_profile_begin/end, bench.py's roofline leg): while armed, every conv launch of
// this host thread carries its own start/stop events, i.e. the kernel's execution time as rocprofv3 reports it
// (an event pair recorded AROUND a launch also counts the dispatch gap, ~3 us).  False when not armed.
// Also true (with *start == nullptr) when a hand-off event is armed (clslam_handoff_arm): the launch then carries it as its
// completion signal.  take_handoff_event(): the same for launches that are never timed (nullptr: none armed).
bool profile_next_events(hipEvent_t* start, hipEvent_t* stop);
hipEvent_t take_handoff_event();

template <typename F, typename Arg>
inline void conv_launch(F kernel, int nblk, hipStream_t stream, const Arg& k) {
#if CLSLAM_DEVICE_BUILD
    hipEvent_t e0, e1;
    if (profile_next_events(&e0, &e1)) {
        hipExtLaunchKernelGGL(kernel, dim3(nblk), dim3(256), 0, stream, e0, e1, 0, k);
        return;
    }
#endif
    hipLaunchKernelGGL(kernel, dim3(nblk), dim3(256), 0, stream, k);
}

}  // namespace clslam
