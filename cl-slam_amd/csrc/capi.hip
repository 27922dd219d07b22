// Library-level entry points: version, thread-local error text, launch checking.
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <utility>
#include <vector>

namespace clslam {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return CLSLAM_ERR_LAUNCH;
    }
    return CLSLAM_OK;
}

// ---- kernel-exact conv timing (measurement only) ---------------------------------------------
static thread_local std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof;
static thread_local int g_prof_cap = 0;   // 0: not armed

bool profile_next_events(hipEvent_t* start, hipEvent_t* stop) {
#if CLSLAM_DEVICE_BUILD
    if (g_prof_cap <= 0 || (int)g_prof.size() >= g_prof_cap) return false;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
    g_prof.emplace_back(a, b);
    *start = a; *stop = b;
    return true;
#else
    (void)start; (void)stop;
    return false;
#endif
}

}  // namespace clslam

extern "C" int clslam_conv_profile_begin(int max_launches) {
    CLSLAM_REQUIRE(max_launches > 0, "conv_profile_begin: max_launches must be positive");
    CLSLAM_REQUIRE(clslam::g_prof_cap == 0, "conv_profile_begin: already armed on this thread");
    clslam::g_prof.clear();
    clslam::g_prof_cap = max_launches;
    return CLSLAM_OK;
}

extern "C" int clslam_conv_profile_end(float* ms, int capacity, int* count) {
    CLSLAM_REQUIRE(count && (ms || capacity == 0), "conv_profile_end: null");
    clslam::g_prof_cap = 0;
    int n = 0, rc = CLSLAM_OK;
#if CLSLAM_DEVICE_BUILD
    for (auto& ev : clslam::g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(ev.second) != hipSuccess || hipEventElapsedTime(&t, ev.first, ev.second) != hipSuccess) {
            clslam::set_error("conv_profile_end: %s", hipGetErrorString(hipGetLastError()));
            rc = CLSLAM_ERR_LAUNCH;
        }
        if (n < capacity) ms[n] = t;
        ++n;
        hipEventDestroy(ev.first); hipEventDestroy(ev.second);
    }
#endif
    clslam::g_prof.clear();
    *count = n;
    return rc;
}

// 101: clslam_conv_desc grew `weight_wino` (appended), double dp_partial in the loss backward entry points
// 102: clslam_conv_desc grew `cu_limit` (appended)
extern "C" int clslam_version(void) { return 102; }
// identity of the kernel sources this library was LINKED from (csrc/build.py passes it when it compiles this file, which it
// does whenever any object is rebuilt): read from the loaded library, not from a file beside it
#ifndef CLSLAM_BUILD_ID
#define CLSLAM_BUILD_ID "unstamped"
#endif
extern "C" const char* clslam_build_id(void) { return CLSLAM_BUILD_ID; }
extern "C" const char* clslam_last_error(void) { return clslam::g_err; }
extern "C" const char* clslam_last_error_string(void) { return clslam::g_err; }      // the name SURVEY.md 8(b) lists
extern "C" int clslam_is_device_build(void) { return CLSLAM_DEVICE_BUILD; }
