// How fast does ONE wave per SIMD issue back-to-back v_mfma_f32_32x32x2_f32, and how much other work of the SAME wave fits
// between two of them for free?  (measurement aid, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NV VALU FMAs and NS SALU adds after every MFMA (kept by asm volatile; independent of the MFMAs)
template <int NACC, int NV, int NS>
__global__ void __launch_bounds__(256) k_probe(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x16{0};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    float v[4] = {a, b, a + b, a - b};
    int sreg = iters;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 3]) : "v"(a), "v"(b));
#pragma unroll
                for (int k = 0; k < NS; ++k) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sreg));
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = v[0] + v[1] + v[2] + v[3] + (float)sreg;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int NS>
void run(float* out, unsigned long long* cyc) {
    const int iters = 200, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k_probe<2, NV, NS><<<grid, 256>>>(out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 16 * 2;
    printf("one wave per SIMD, 2 accumulators, %2d VALU + %2d SALU after every MFMA: %.2f ns per MFMA\n", NV, NS, ms * 1e6 / n);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
    run<0, 0>(out, cyc); run<4, 0>(out, cyc); run<8, 0>(out, cyc); run<12, 0>(out, cyc); run<16, 0>(out, cyc); run<24, 0>(out, cyc);
    run<0, 8>(out, cyc); run<0, 16>(out, cyc); run<8, 8>(out, cyc); run<12, 12>(out, cyc);
    return 0;
}
