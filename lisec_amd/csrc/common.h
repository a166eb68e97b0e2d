// Shared host/device helpers for the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/lisec_hip.h"

namespace lisec {

constexpr int kWave = 64;  // CDNA4 wavefront

void set_error(const char* fmt, ...);

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
