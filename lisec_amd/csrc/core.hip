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

static lisec_tuning g_tuning = {(int)sizeof(lisec_tuning), 12, 3, 2, 1, 1, 1, -1, 0, 0, 1024, 0, 0, 32, 1024, 1, 2, 1, 0, 0};
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
    LISEC_CHECK_ARG((t->wgrad_per_cu == 2 || t->wgrad_per_cu == 3) && t->wgrad_combine_max >= 0 && t->wgrad_combine_max <= 4096 &&
                    (t->lone_db == 0 || t->lone_db == 1) && (t->wgrad_ring == 0 || t->wgrad_ring == 1) &&
                    (t->wide_tile == 0 || t->wide_tile == 1) && t->wgrad_ring_slots >= 0 && t->force_splitk >= 0 && (t->plane_pair == 0 || t->plane_pair == 1) &&
                    (t->dense64 == 0 || t->dense64 == 1) && (t->half_n == 0 || t->half_n == 1) && t->field_seg >= 0 &&
                    t->field_tpw >= 0 && t->vfe_shape >= -1,
                    "tuning: wgrad_per_cu 2|3, wgrad_combine_max 0..4096, flags 0|1, wgrad_ring_slots / force_splitk / "
                    "field_seg / field_tpw >= 0, vfe_shape >= -1");
    lisec::g_tuning = *t;
    return LISEC_OK;
}

// Workspaces whose head holds arrival counters (lisec_conv_forward* with K slices, lisec_conv_wgrad*) must start zero-filled;
// every call leaves the counters at zero.  This is that first fill, for callers that do not have a memset of their own.
extern "C" int lisec_workspace_init(void* workspace, size_t bytes, lisec_stream_t stream) {
    LISEC_CHECK_ARG(workspace || bytes == 0, "NULL workspace");
    if (bytes) LISEC_HIP_TRY(hipMemsetAsync(workspace, 0, bytes, static_cast<hipStream_t>(stream)));
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
