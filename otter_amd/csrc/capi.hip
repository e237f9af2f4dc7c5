// Error plumbing and version query of the C ABI.
#include "common.hpp"
#include "otter_hip.h"
#include <stdarg.h>

namespace {
thread_local char g_error[512] = "";
}

void otter_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

extern "C" const char* otter_last_error(void) { return g_error; }
extern "C" int otter_hip_abi_version(void) { return OTTER_HIP_ABI_VERSION; }
