// Error plumbing and version query of the C ABI.
#include "common.hpp"
#include "otter_hip.h"
#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>

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

// Dynamic-LDS opt-in is a per-device property of a kernel: remember the largest size granted per (kernel, device), so a
// process that drives several GPUs (or several threads launching at once) configures each of them.
int otter_reserve_lds(const void* kernel, size_t bytes, const char* name) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        otter_set_error("%s: no current HIP device", name);
        return OTTER_ERR_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[{kernel, dev}];
    if (bytes <= have) return OTTER_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        otter_set_error("%s: cannot reserve %zu bytes of LDS: %s", name, bytes, hipGetErrorString(e));
        return OTTER_ERR_LAUNCH;
    }
    have = bytes;
    return OTTER_OK;
}

// Compute units of the current device (cached per device); 0 with the error set when the query fails.
int otter_cu_count(const char* name) {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        otter_set_error("%s: no current HIP device", name);
        return 0;
    }
    std::lock_guard<std::mutex> lock(mu);
    int& n = cus[dev];
    if (n == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            otter_set_error("%s: cannot query the device", name);
            return 0;
        }
        n = prop.multiProcessorCount;
    }
    return n;
}
