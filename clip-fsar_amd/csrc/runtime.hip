// Error plumbing + version for libclipfsar_hip (no global device state lives in this library).
#include <stdarg.h>

#include "common.h"

thread_local char cfsar_err_buf[512] = {0};

int cfsar_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(cfsar_err_buf, sizeof(cfsar_err_buf), fmt, ap);
    va_end(ap);
    return 1;
}

int cfsar_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfsar_fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

extern "C" int cfsar_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* cfsar_last_error(void) { return cfsar_err_buf; }
