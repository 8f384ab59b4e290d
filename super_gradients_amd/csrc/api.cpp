// Error plumbing and version of libsgx_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/sgx_hip.h"

static thread_local char g_err[512] = "";

void sgx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sgx_last_error(void) { return g_err; }
extern "C" int32_t sgx_version(void) { return 100; /* 0.1.0 */ }
