// Error reporting + device queries of the C ABI (include/lisec_hip.h).
#include "common.h"

#include <cstring>

namespace lisec {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace lisec

extern "C" const char* lisec_last_error(void) { return lisec::g_err; }

extern "C" int lisec_abi_version(void) { return 9; }

extern "C" int lisec_device_info(char* name, int cap) {
    int dev = 0;
    LISEC_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    LISEC_HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name && cap > 0) {
        strncpy(name, p.gcnArchName, cap - 1);
        name[cap - 1] = 0;
    }
    return p.multiProcessorCount;
}
