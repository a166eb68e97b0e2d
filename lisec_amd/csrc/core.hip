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

static lisec_tuning g_tuning = {(int)sizeof(lisec_tuning), 12, 3, 2, 1, 1, 1, -1, 0, 0, 1024, 0, 0, 32, 1024, 1, 2, 1, 0};
const lisec_tuning& tuning() { return g_tuning; }
}  // namespace lisec

extern "C" int lisec_tuning_get(lisec_tuning* t) {
    LISEC_CHECK_ARG(t, "NULL tuning record");
    *t = lisec::g_tuning;
    return LISEC_OK;
}

extern "C" int lisec_tuning_set(const lisec_tuning* t) {
    LISEC_CHECK_ARG(t && t->struct_bytes == (int)sizeof(lisec_tuning), "tuning record of another ABI version");
    LISEC_CHECK_ARG(t->max_splitk >= 1 && t->splitk_min_steps >= 1 && t->min_splitk >= 2 && t->wgrad_blocks >= 1 && t->wgrad_batch_blocks >= 1,
                    "tuning: max_splitk, splitk_min_steps, wgrad_blocks >= 1, min_splitk >= 2");
    lisec::g_tuning = *t;
    return LISEC_OK;
}

extern "C" const char* lisec_last_error(void) { return lisec::g_err; }

extern "C" int lisec_abi_version(void) { return 10; }

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
