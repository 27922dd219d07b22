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

// the hand-off event armed by clslam_handoff_arm, waiting for the next launch that can carry it
static thread_local hipEvent_t g_handoff = nullptr;

hipEvent_t take_handoff_event() {
    hipEvent_t e = g_handoff;
    g_handoff = nullptr;
    return e;
}

bool profile_next_events(hipEvent_t* start, hipEvent_t* stop) {
#if CLSLAM_DEVICE_BUILD
    if (g_prof_cap <= 0 || (int)g_prof.size() >= g_prof_cap) {
        // not measuring: an armed hand-off event rides on this launch as its completion signal (no start event)
        hipEvent_t h = take_handoff_event();
        if (h == nullptr) return false;
        *start = nullptr; *stop = h;
        return true;
    }
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

extern "C" void* clslam_handoff_event_create(void) {
#if CLSLAM_DEVICE_BUILD
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        clslam::set_error("handoff_event_create: %s", hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    return (void*)e;
#else
    return nullptr;      // the emulator has one stream: nothing to hand off
#endif
}

extern "C" void clslam_handoff_event_destroy(void* event) {
#if CLSLAM_DEVICE_BUILD
    if (event) (void)hipEventDestroy((hipEvent_t)event);
#else
    (void)event;
#endif
}

extern "C" int clslam_handoff_arm(void* event) {
    CLSLAM_REQUIRE(event, "handoff_arm: null event");
    // An event still armed here was armed for a launch that never went out (the call between arm and wait failed its argument
    // checks) and was never recorded: it is dropped, nobody waits for it.  Refusing instead left the host thread
    // unable to arm ever again after one recoverable error.
    clslam::g_handoff = (hipEvent_t)event;
    return CLSLAM_OK;
}

extern "C" int clslam_handoff_wait(void* event, void* producer_stream, void* consumer_stream) {
    CLSLAM_REQUIRE(event, "handoff_wait: null event");
#if CLSLAM_DEVICE_BUILD
    hipEvent_t e = (hipEvent_t)event;
    if (clslam::g_handoff == e) {       // nothing launched since the arm call could carry it: an ordinary record
        clslam::g_handoff = nullptr;
        if (hipEventRecord(e, (hipStream_t)producer_stream) != hipSuccess) {
            clslam::set_error("handoff_wait: hipEventRecord: %s", hipGetErrorString(hipGetLastError()));
            return CLSLAM_ERR_LAUNCH;
        }
    }
    if (hipStreamWaitEvent((hipStream_t)consumer_stream, e, 0) != hipSuccess) {
        clslam::set_error("handoff_wait: hipStreamWaitEvent: %s", hipGetErrorString(hipGetLastError()));
        return CLSLAM_ERR_LAUNCH;
    }
#else
    (void)producer_stream; (void)consumer_stream;
    if (clslam::g_handoff == (hipEvent_t)event) clslam::g_handoff = nullptr;
#endif
    return CLSLAM_OK;
}

// 101: clslam_conv_desc grew `weight_wino` (appended), double dp_partial in the loss backward entry points
// 102: clslam_conv_desc grew `cu_limit` (appended)
// 103: clslam_handoff_* (additive)
// 104: clslam_*_pyramid_range (additive)
extern "C" int clslam_version(void) { return 104; }
// identity of the kernel sources this library was LINKED from (csrc/build.py passes it when it compiles this file, which it
// does whenever any object is rebuilt): read from the loaded library, not from a file beside it
#ifndef CLSLAM_BUILD_ID
#define CLSLAM_BUILD_ID "unstamped"
#endif
extern "C" const char* clslam_build_id(void) { return CLSLAM_BUILD_ID; }
extern "C" const char* clslam_last_error(void) { return clslam::g_err; }
extern "C" const char* clslam_last_error_string(void) { return clslam::g_err; }      // the name SURVEY.md 8(b) lists
extern "C" int clslam_is_device_build(void) { return CLSLAM_DEVICE_BUILD; }
