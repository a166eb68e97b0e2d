// What does a sustained fp32-MFMA loop really get on this chip?  (measurement aid, not part of the library)
// Every SIMD of every CU issues back-to-back MFMAs on random operands held in registers; the loop is bracketed by
// s_memtime (shader cycles) and s_memrealtime (100 MHz) stamps, so the in-kernel clock, the cycles per MFMA and the
// nanoseconds per MFMA come from the same launch.  Shapes: v_mfma_f32_32x32x2_f32 (64 cycles, 4096 FLOP) and
// v_mfma_f32_16x16x4_f32 (32 cycles, 2048 FLOP); 1, 2 and 3 waves per SIMD.  The chip is warmed for ~2 s first (DVFS).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_rate tools/probe/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC>
__global__ void __launch_bounds__(1024) k_rate(const float* __restrict__ src, float* out, unsigned long long* stamps, int iters) {
    const float a = src[threadIdx.x], b = src[threadIdx.x + 1024];
    float sum = 0.f;
    unsigned long long c0, c1, r0, r1;
    if (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x16{0};
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    } else {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0};
        c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        c1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) sum += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int SHAPE, int NACC>
static void run(const char* name, int waves_per_simd, const float* src, float* out, unsigned long long* stamps, int iters) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, src, out, stamps, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_rate<SHAPE, NACC>), dim3(blocks), dim3(threads), 0, 0, src, out, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const int nw = blocks * threads / 64;
    std::vector<unsigned long long> h(2 * nw);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
    std::vector<double> clk(nw), cyc(nw);
    const double mfma_per_wave = (double)iters * 16 * NACC;
    for (int i = 0; i < nw; ++i) {
        clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);       // cycles per ns = GHz
        cyc[i] = (double)h[2 * i] / mfma_per_wave * 1.0;                  // cycles of this wave's loop per own MFMA
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double flop = (SHAPE == 32 ? 4096.0 : 2048.0) * mfma_per_wave * nw;
    const double tf = flop / (ms * 1e-3) / 1e12;
    const double ns = ms * 1e6 / (mfma_per_wave * waves_per_simd);       // wall ns per MFMA of one SIMD
    printf("%-28s %d wave(s)/SIMD  %8.3f ms  %6.1f TFLOP/s (%.3f of 157.3)  %6.2f ns per MFMA per SIMD  in-kernel clock %.3f GHz (p10 %.3f p90 %.3f)  "
           "%.1f cycles per own MFMA\n", name, waves_per_simd, ms, tf, tf / 157.3, ns, clk[nw / 2], clk[nw / 10], clk[nw * 9 / 10], cyc[nw / 2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float *src, *out;
    unsigned long long* stamps;
    hipMalloc(&src, 2048 * sizeof(float));
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&stamps, 2 * 256 * 16 * sizeof(unsigned long long));
    std::vector<float> h(2048);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(src, h.data(), sizeof(float) * 2048, hipMemcpyHostToDevice);
    // ~2 s of back-to-back launches so that the chip settles at the clock it holds under this load
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float total = 0;
        while (total < 2000.f) {
            hipEventRecord(e0);
            for (int w = 0; w < 20; ++w) hipLaunchKernelGGL((k_rate<32, 4>), dim3(256), dim3(256), 0, 0, src, out, stamps, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            total += ms;
        }
    }
    run<32, 1>("32x32x2, 1 accumulator", 1, src, out, stamps, iters);
    run<32, 2>("32x32x2, 2 accumulators", 1, src, out, stamps, iters);
    run<32, 4>("32x32x2, 4 accumulators", 1, src, out, stamps, iters);
    run<32, 3>("32x32x2, 3 accumulators", 2, src, out, stamps, iters);
    run<32, 3>("32x32x2, 3 accumulators", 3, src, out, stamps, iters);
    run<32, 4>("32x32x2, 4 accumulators", 4, src, out, stamps, iters);
    run<16, 4>("16x16x4, 4 accumulators", 1, src, out, stamps, iters);
    run<16, 8>("16x16x4, 8 accumulators", 1, src, out, stamps, iters);
    run<16, 8>("16x16x4, 8 accumulators", 2, src, out, stamps, iters);
    run<16, 8>("16x16x4, 8 accumulators", 3, src, out, stamps, iters);
    run<32, 4>("32x32x2, 4 accumulators", 1, src, out, stamps, iters);   // again: the first line after the others
    return 0;
}
