// Host-side runtime bits of libhh: version, thread-local error string, launch check.
#include "common.h"

static thread_local char g_err[512] = "";

void hh_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hh_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hh_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return HH_ERR_LAUNCH;
    }
    return HH_OK;
}

extern "C" int hh_version(void) { return 100; }
extern "C" const char* hh_last_error_string(void) { return g_err; }
