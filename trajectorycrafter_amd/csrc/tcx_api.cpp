// Library-level entry points of libtcx_hip.so: version, device info, thread-local error string.
#include <stdarg.h>
#include <string.h>
#include "tcx_common.h"

static thread_local char g_err[512] = "";

void tcx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int tcx_version(void) { return 1; }

extern "C" const char* tcx_last_error_string(void) { return g_err; }

extern "C" int tcx_device_info(int device, int32_t out[4]) {
    TCX_CHECK(out != nullptr, TCX_E_NULL, "tcx_device_info: out is null");
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        tcx_set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
        return (int)e;
    }
    out[0] = prop.multiProcessorCount;
    out[1] = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
    out[2] = prop.warpSize;
    int arch = 0;
    const char* g = strstr(prop.gcnArchName, "gfx");
    if (g) arch = atoi(g + 3);
    out[3] = arch;
    return TCX_OK;
}

uint32_t tcx_cu_count() {
    static std::atomic<uint32_t> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    uint32_t n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = (uint32_t)v;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
