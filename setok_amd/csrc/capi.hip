// capi.hip — error plumbing and device query of the C ABI (include/setok_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

char* setok_err_buf() { return g_err; }

int setok_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int setok_abi_version(void) { return SETOK_ABI_VERSION; }
extern "C" const char* setok_last_error(void) { return g_err; }

extern "C" int setok_device_info(char* name_host, int name_cap, int* cu_count_host) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        if (name_host && name_cap > 0) name_host[0] = 0;
        if (cu_count_host) *cu_count_host = 0;
        return setok_fail(SETOK_EUNSUPPORTED, "no HIP device visible");
    }
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return setok_fail(SETOK_ELAUNCH, "hipGetDeviceProperties failed");
    if (name_host && name_cap > 0) {
        snprintf(name_host, name_cap, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count_host) *cu_count_host = p.multiProcessorCount;
    return SETOK_OK;
}
