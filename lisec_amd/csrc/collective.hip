// Data-parallel gradient exchange of the C ABI: RCCL all-reduce (sum) of the flat fp32 gradient over xGMI,
// then the 1/world scale.  The reference has no distributed code (model_training.py:299 is one process);
// whole samples per GPU, ONE exchange per step.
//
// RCCL is resolved at run time (dlopen by soname, preferring the copy the process has ALREADY loaded --
// PyTorch ships its own librccl.so.1, and a communicator must be used with the library that made it), so
// the shared library has no link-time dependency on it and loads on a machine without RCCL; the entry
// points then fail with LISEC_EHIP and a message.
#include "common.h"

#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

namespace lisec {
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "symbols missing";      // the loader's message, taken when dlopen failed (dlerror() clears itself)
};

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)                                   // the copy already in the process first
            if ((x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!x.handle) {
            for (const char* n : names) {
                if ((x.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
                if (const char* e = dlerror()) std::snprintf(x.why, sizeof(x.why), "%s", e);
            }
        }
        if (!x.handle) return x;
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(x.handle, "ncclCommCount"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(x.handle, "ncclAllReduce"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.CommCount && x.AllReduce && x.GetErrorString;
        return x;
    }();
    return r;
}

#define LISEC_RCCL_TRY(expr)                                                                  \
    do {                                                                                      \
        ncclResult_t r__ = (expr);                                                            \
        if (r__ != ncclSuccess) {                                                             \
            ::lisec::set_error("%s failed: %s", #expr, rccl().GetErrorString(r__));           \
            return LISEC_EHIP;                                                                \
        }                                                                                     \
    } while (0)

int need_rccl() {
    if (!rccl().ok) {
        set_error("RCCL (librccl.so.1) could not be loaded: %s", rccl().why);
        return LISEC_EHIP;
    }
    return LISEC_OK;
}

}  // namespace
}  // namespace lisec

using namespace lisec;

static_assert(LISEC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "lisec_comm id size must match ncclUniqueId");

extern "C" int lisec_comm_unique_id(void* id) {
    LISEC_CHECK_ARG(id, "NULL id");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId u;
    LISEC_RCCL_TRY(rccl().GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return LISEC_OK;
}

extern "C" int lisec_comm_probe(void) { return need_rccl(); }

extern "C" int lisec_comm_count(lisec_comm_t comm, int* count) {
    LISEC_CHECK_ARG(comm && count, "NULL communicator / count");
    if (int rc = need_rccl()) return rc;
    LISEC_RCCL_TRY(rccl().CommCount(static_cast<ncclComm_t>(comm), count));
    return LISEC_OK;
}

extern "C" int lisec_comm_init(int rank, int world, const void* id, lisec_comm_t* comm) {
    LISEC_CHECK_ARG(id && comm && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    LISEC_RCCL_TRY(rccl().CommInitRank(&c, world, u, rank));
    *comm = c;
    return LISEC_OK;
}

extern "C" int lisec_comm_destroy(lisec_comm_t comm) {
    if (!comm) return LISEC_OK;
    if (int rc = need_rccl()) return rc;
    LISEC_RCCL_TRY(rccl().CommDestroy(static_cast<ncclComm_t>(comm)));
    return LISEC_OK;
}

extern "C" int lisec_allreduce_grads(lisec_comm_t comm, float* grad, long long n, int world, lisec_stream_t stream) {
    LISEC_CHECK_ARG(comm && grad && n >= 0 && n % 4 == 0, "bad all-reduce arguments (n must be a multiple of 4)");
    if (int rc = need_rccl()) return rc;
    // the divisor: the communicator's own size; a caller-supplied world that disagrees with it would silently mis-scale
    // every gradient, so it is refused -- a deliberate other divisor is passed as -divisor
    int count = 0;
    LISEC_RCCL_TRY(rccl().CommCount(static_cast<ncclComm_t>(comm), &count));
    LISEC_CHECK_ARG(world <= 0 || world == count, "world = %d but the communicator has %d ranks (0 = use the communicator's "
                    "size, -d = explicit divisor d)", world, count);
    world = world < 0 ? -world : count;
    if (n == 0) return LISEC_OK;
    if (StepPlan* plan = plan_recording()) {
        // part of a recorded step (lisec_step_plan_*): the collective is re-issued at this place of the sequence by every
        // replay -- all ranks record and replay the same schedule, so RCCL sees the same order of collectives everywhere
        auto fn = rccl().AllReduce;
        const ncclComm_t c = static_cast<ncclComm_t>(comm);
        const hipStream_t st = static_cast<hipStream_t>(stream);
        plan_append(plan, [=]() { return fn(grad, grad, (size_t)n, ncclFloat, ncclSum, c, st) == ncclSuccess ? hipSuccess : hipErrorUnknown; });
    }
    LISEC_RCCL_TRY(rccl().AllReduce(grad, grad, (size_t)n, ncclFloat, ncclSum, static_cast<ncclComm_t>(comm),
                                    static_cast<hipStream_t>(stream)));
    if (world > 1) return lisec_scale(grad, n, 1.0f / (float)world, stream);
    return LISEC_OK;
}
