// Shared host/device helpers for the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <functional>
#include <tuple>
#include <utility>

#include "../../include/lisec_hip.h"

namespace lisec {

constexpr int kWave = 64;  // CDNA4 wavefront

void set_error(const char* fmt, ...);
const lisec_tuning& tuning();      // the process-wide launch-plan knobs (core.hip; lisec_tuning_set)

#define LISEC_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            ::lisec::set_error(__VA_ARGS__);       \
            return LISEC_EINVAL;                   \
        }                                          \
    } while (0)

#define LISEC_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::lisec::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),   \
                               __FILE__, __LINE__);                                      \
            return LISEC_EHIP;                                                           \
        }                                                                                \
    } while (0)

#define LISEC_LAUNCH_CHECK()                                                              \
    do {                                                                                  \
        hipError_t e__ = hipGetLastError();                                               \
        if (e__ != hipSuccess) {                                                          \
            ::lisec::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                               __FILE__, __LINE__);                                       \
            return LISEC_EHIP;                                                            \
        }                                                                                 \
    } while (0)

// ---- step plans (plan.hip; lisec_step_plan_* of the ABI) -----------------------------------------------------------------
// Every kernel launch of the library goes through lisec::launch.  While the calling thread records a step plan, the
// launch (kernel, geometry, stream and a COPY of its arguments) is also appended to the plan, so that lisec_step_plan_run
// can re-issue the whole step -- ~250 launches on two streams with their fork / join events -- from one C call, without
// the Python schedule, the plan selection or the argument marshalling that produced it.
struct StepPlan;
StepPlan* plan_recording();                                   // the plan this thread is recording, or nullptr
void plan_append(StepPlan* plan, std::function<hipError_t()> op);

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    if (StepPlan* plan = plan_recording()) {
        std::tuple<KArgs...> params{static_cast<KArgs>(args)...};
        plan_append(plan, [=]() -> hipError_t {
            std::apply([&](const KArgs&... a) { hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, a...); }, params);
            return hipGetLastError();
        });
    }
    hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, static_cast<KArgs>(args)...);
}
#define LISEC_LAUNCH(...) ::lisec::launch(__VA_ARGS__)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bump allocator over a caller-provided workspace (all carve-outs 256 B aligned).
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t n) {
        T* r = reinterpret_cast<T*>(base + off);
        off = align_up(off + n * sizeof(T), 256);
        return r;
    }
};

#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Order-independent (hence deterministic) cross-workgroup sums of doubles: a value is split into two integer limbs
// (units of 2^-8 and 2^-40) that are added with 64-bit integer atomics; integer addition commutes, so the total does not
// depend on the arrival order.  Resolution 2^-40 per addend; exact range |total| < 2^45, graceful beyond.
__device__ __forceinline__ void fx_split(double s, long long& hi, long long& lo) {
    double h = floor(s * 256.0);
    h = fmin(fmax(h, -9.0e18), 9.0e18);
    hi = (long long)h;
    lo = (long long)rint((s - h * (1.0 / 256.0)) * 1099511627776.0);
}
__device__ __forceinline__ double fx_join(long long hi, long long lo) {
    return (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 1099511627776.0);
}
__device__ __forceinline__ void fx_atomic_add(long long* acc, double s) {      // acc[0] = hi limb, acc[1] = lo limb
    long long hi, lo;
    fx_split(s, hi, lo);
    atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)hi);
    atomicAdd(reinterpret_cast<unsigned long long*>(acc + 1), (unsigned long long)lo);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T u = __shfl_xor(v, o, 64);
        v = u < v ? u : v;
    }
    return v;
}
#endif

}  // namespace lisec
