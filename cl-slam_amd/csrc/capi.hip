// Library-level entry points: version, thread-local error text, launch checking.
#include "common.h"

#include <cstdarg>
#include <cstdio>

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

}  // namespace clslam

extern "C" int clslam_version(void) { return 100; }
extern "C" const char* clslam_last_error(void) { return clslam::g_err; }
extern "C" int clslam_is_device_build(void) { return CLSLAM_DEVICE_BUILD; }
