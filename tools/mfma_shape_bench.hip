// Micro-benchmark: LDS-read + fp32-MFMA inner loops of the implicit-GEMM kernel (csrc/igemm.hip) on random data,
// for the two f32 MFMA shapes of gfx950 and two wave tiles.  No global traffic inside the loop: this is the
// ceiling of the inner loop alone (the chip lowers its clock under MFMA load, so shapes are ranked by WALL time).
//   A  32x32x2, wave tile 32x64 (2 accumulators)   -- what k_igemm does today
//   B  32x32x2, wave tile 64x64 (4 accumulators)
//   C  16x16x4, wave tile 32x64 (8 accumulators)
//   D  16x16x4, wave tile 64x64 (16 accumulators)
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_bench.hip -o tools/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 64;

__device__ __forceinline__ float hash01(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(x & 0xffffff) / 16777216.f - 0.5f;
}

// 64-row wave tiles stage 256 rows per block: with BK = 32 (LDA 36) the block still fits three times per CU
template <int VARIANT>
__global__ void __launch_bounds__(256) k_loop(float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = (VARIANT == 0 || VARIANT == 2) ? 128 : 256;
    constexpr int BK = (VARIANT == 0 || VARIANT == 2) ? 64 : 32;
    constexpr int LDA = BK + 4;
    float* sA = smem;
    float* sB = smem + ROWS * LDA;
    for (int i = threadIdx.x; i < ROWS * LDA + BK * BN; i += 256) smem[i] = hash01(i * 2654435761u + blockIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc_sum = 0.f;
    if (VARIANT == 0) {
        f32x16 acc0 = {0}, acc1 = {0};
        const float* aRow = sA + (wave * 32 + (lane & 31)) * LDA + 4 * (lane >> 5);
        const float* bCol = sB + ((lane >> 5) * BN + (lane & 31)) * 4;
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");          // the fragments are re-read from LDS every iteration
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                const float4 a = *reinterpret_cast<const float4*>(aRow + kc * 8);
                const float4 b0 = *reinterpret_cast<const float4*>(bCol + kc * 2 * BN * 4);
                const float4 b1 = *reinterpret_cast<const float4*>(bCol + kc * 2 * BN * 4 + 32 * 4);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) acc_sum += acc0[r] + acc1[r];
    } else if (VARIANT == 1) {
        f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
        const float* aRow = sA + (wave * 64 + (lane & 31)) * LDA + 4 * (lane >> 5);
        const float* bCol = sB + ((lane >> 5) * BN + (lane & 31)) * 4;
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");          // the fragments are re-read from LDS every iteration
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                const float4 a0 = *reinterpret_cast<const float4*>(aRow + kc * 8);
                const float4 a1 = *reinterpret_cast<const float4*>(aRow + 32 * LDA + kc * 8);
                const float4 b0 = *reinterpret_cast<const float4*>(bCol + kc * 2 * BN * 4);
                const float4 b1 = *reinterpret_cast<const float4*>(bCol + kc * 2 * BN * 4 + 32 * 4);
#define STEP(X)                                                                   \
                c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X, b0.X, c00, 0, 0, 0); \
                c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X, b1.X, c01, 0, 0, 0); \
                c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X, b0.X, c10, 0, 0, 0); \
                c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X, b1.X, c11, 0, 0, 0);
                STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
            }
        }
        for (int r = 0; r < 16; ++r) acc_sum += c00[r] + c01[r] + c10[r] + c11[r];
    } else {
        constexpr int MI = VARIANT == 2 ? 2 : 4;         // 16-row tiles per wave
        constexpr int WROWS = MI * 16;
        f32x4 c[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = f32x4{0, 0, 0, 0};
        // lane: row/col = lane & 15, k quad = lane >> 4
        const float* aRow = sA + (wave * WROWS + (lane & 15)) * LDA + 4 * (lane >> 4);
        const float* bCol = sB + ((lane >> 4) * BN + (lane & 15)) * 4;     // packed [k/4][n][4]
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");          // the fragments are re-read from LDS every iteration
#pragma unroll
            for (int kc = 0; kc < BK / 16; ++kc) {
                float4 a[MI], b[4];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const float4*>(aRow + i * 16 * LDA + kc * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float4*>(bCol + kc * 4 * BN * 4 + j * 16 * 4);
#define STEP(X)                                                                                     \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].X, b[j].X, c[i][j], 0, 0, 0);
                STEP(x) STEP(y) STEP(z) STEP(w)
#undef STEP
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc_sum += c[i][j][0] + c[i][j][1] + c[i][j][2] + c[i][j][3];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc_sum;
}


// Variant A plus, step by step, what k_igemm does around the MFMAs of one 64-deep K slab:
//   STAGE 1: two __syncthreads per slab      2: + 12 ds_write_b128 per thread between them (register -> LDS)
//   STAGE 3: + 12 global float4 loads per thread per slab (issued before the MFMAs, consumed by the LDS writes)
//   STAGE 4: as 3 with the software-pipelined fragment reads + sched_barrier of k_igemm
template <int STAGE>
__global__ void __launch_bounds__(256) k_staged(float* __restrict__ out, const float* __restrict__ src, size_t src_floats,
                                                int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BK = 64, LDA = 68;
    float* sA = smem;
    float* sB = smem + 128 * LDA;
    for (int i = threadIdx.x; i < 128 * LDA + BK * BN; i += 256) smem[i] = hash01(i * 2654435761u + blockIdx.x);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, piece = tid & 15;
    f32x16 acc0 = {0}, acc1 = {0};
    const float* aRow = sA + (wave * 32 + (lane & 31)) * LDA + 4 * (lane >> 5);
    const float* bCol = sB + ((lane >> 5) * BN + (lane & 31)) * 4;
    float4 ra[8], rb[4];
#pragma unroll
    for (int p = 0; p < 8; ++p) ra[p] = make_float4(hash01(tid + p), hash01(tid + 2 * p), 0.25f, -0.5f);
#pragma unroll
    for (int p = 0; p < 4; ++p) rb[p] = make_float4(hash01(tid + 7 * p), 0.125f, hash01(tid + 3 * p), 0.5f);
    // tile-like global pattern: 128 rows of 64 floats (256 B each), row r at (base + r) * 64, base walks the buffer
    size_t base = ((size_t)blockIdx.x * 128) % (src_floats / 64 - 4096);
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        const bool stageA = STAGE != 5 || (it % 3) == 2;     // STAGE 5: one A tile serves the three kw taps
        if (STAGE >= 3) {
            if (stageA)
#pragma unroll
            for (int p = 0; p < 8; ++p)
                ra[p] = *reinterpret_cast<const float4*>(src + (base + p * 16 + (tid >> 4)) * 64 + piece * 4);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                rb[p] = *reinterpret_cast<const float4*>(src + (size_t)(it % 27) * 4096 + p * 1024 + tid * 4);
            base += 400;                                   // the next tap: one image row further
            if (base + 4096 > src_floats / 64) base = (size_t)blockIdx.x * 128;
        }
        if (STAGE >= 4) {
            float4 a = *reinterpret_cast<const float4*>(aRow);
            float4 b0 = *reinterpret_cast<const float4*>(bCol);
            float4 b1 = *reinterpret_cast<const float4*>(bCol + 32 * 4);
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                float4 an = a, b0n = b0, b1n = b1;
                if (kc + 1 < BK / 8) {
                    an = *reinterpret_cast<const float4*>(aRow + (kc + 1) * 8);
                    b0n = *reinterpret_cast<const float4*>(bCol + (kc + 1) * 2 * BN * 4);
                    b1n = *reinterpret_cast<const float4*>(bCol + (kc + 1) * 2 * BN * 4 + 32 * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a = an; b0 = b0n; b1 = b1n;
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < BK / 8; ++kc) {
                const float4 a = *reinterpret_cast<const float4*>(aRow + kc * 8);
                const float4 b0 = *reinterpret_cast<const float4*>(bCol + kc * 2 * BN * 4);
                const float4 b1 = *reinterpret_cast<const float4*>(bCol + kc * 2 * BN * 4 + 32 * 4);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            }
        }
        if (STAGE >= 1) __syncthreads();
        if (STAGE >= 2) {
            if (stageA)
#pragma unroll
            for (int p = 0; p < 8; ++p)
                *reinterpret_cast<float4*>(sA + (p * 16 + (tid >> 4)) * LDA + piece * 4) = ra[p];
            float* bl = sB + ((tid >> 6) * BN + (tid & 63)) * 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(bl + p * 4 * BN * 4) = rb[p];
        }
        if (STAGE >= 1) __syncthreads();
    }
    float acc_sum = 0.f;
    for (int r = 0; r < 16; ++r) acc_sum += acc0[r] + acc1[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc_sum;
}

template <int STAGE>
double run_staged(const char* name, int blocks, int iters, float* out, const float* src, size_t src_floats) {
    const size_t lds = (size_t)(128 * 68 + 64 * BN) * sizeof(float);
    hipFuncSetAttribute((const void*)k_staged<STAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_staged<STAGE>, dim3(blocks), dim3(256), lds, 0, out, src, src_floats, iters);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_staged<STAGE>, dim3(blocks), dim3(256), lds, 0, out, src, src_floats, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = (double)blocks * iters * 4 * 32 * 64 * 64 * 2;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-46s blocks %4d  %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", name, blocks, ms, tf, tf / 157.3);
    return tf;
}

// S5: the same work as S3/S4 restructured for intra-wave overlap: K slab of 32 (LDA 36), LDS double buffer, ONE
// barrier per slab; the registers of slab s+1 are written to the other buffer and refilled with slab s+2 by global
// loads BETWEEN the MFMAs of slab s (one companion group after every 4 MFMAs, pinned with sched_barrier).
__global__ void __launch_bounds__(256) k_dbuf(float* __restrict__ out, const float* __restrict__ src, size_t src_floats,
                                              int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BK = 32, LDA = 36, A_FL = 128 * LDA, B_FL = BK * BN, BUF = A_FL + B_FL;
    for (int i = threadIdx.x; i < 2 * BUF; i += 256) smem[i] = hash01(i * 2654435761u + blockIdx.x);
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, piece = tid & 7;
    f32x16 acc0 = {0}, acc1 = {0};
    float4 ra[4], rb[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = make_float4(hash01(tid + p), hash01(tid + 2 * p), 0.25f, -0.5f);
#pragma unroll
    for (int p = 0; p < 2; ++p) rb[p] = make_float4(hash01(tid + 7 * p), 0.125f, hash01(tid + 3 * p), 0.5f);
    size_t base = ((size_t)blockIdx.x * 128) % (src_floats / 64 - 4096);
    // staging map: 128 rows x 8 pieces of 16 B: row = p*32 + tid/8, piece = tid%8
    for (int it = 0; it < iters; ++it) {
        float* cur = smem + (it & 1) * BUF;
        float* nxt = smem + ((it + 1) & 1) * BUF;
        const float* aRow = cur + (wave * 32 + (lane & 31)) * LDA + 4 * (lane >> 5);
        const float* bCol = cur + A_FL + ((lane >> 5) * BN + (lane & 31)) * 4;
        const int half = it & 1;                         // two 32-slabs = one 64-channel tap
        float4 a = *reinterpret_cast<const float4*>(aRow);
        float4 b0 = *reinterpret_cast<const float4*>(bCol);
        float4 b1 = *reinterpret_cast<const float4*>(bCol + 32 * 4);
#pragma unroll
        for (int kc = 0; kc < BK / 8; ++kc) {
            float4 an = a, b0n = b0, b1n = b1;
            if (kc + 1 < BK / 8) {
                an = *reinterpret_cast<const float4*>(aRow + (kc + 1) * 8);
                b0n = *reinterpret_cast<const float4*>(bCol + (kc + 1) * 2 * BN * 4);
                b1n = *reinterpret_cast<const float4*>(bCol + (kc + 1) * 2 * BN * 4 + 32 * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // companion group 2*kc: stage piece kc of A (write regs of slab s+1, then refill them with slab s+2)
            *reinterpret_cast<float4*>(nxt + (kc * 32 + (tid >> 3)) * LDA + piece * 4) = ra[kc];
            ra[kc] = *reinterpret_cast<const float4*>(src + (base + kc * 32 + (tid >> 3)) * 64 + half * 32 + piece * 4);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kc < 2) {                                // companion group 2*kc+1: the weight slab (2 pieces)
                *reinterpret_cast<float4*>(nxt + A_FL + (kc * 1024 + tid * 4)) = rb[kc];
                rb[kc] = *reinterpret_cast<const float4*>(src + (size_t)((it >> 1) % 27) * 4096 + half * 2048 + kc * 1024 + tid * 4);
            }
            a = an; b0 = b0n; b1 = b1n;
        }
        if (half) { base += 400; if (base + 4096 > src_floats / 64) base = (size_t)blockIdx.x * 128; }
        __syncthreads();
    }
    float acc_sum = 0.f;
    for (int r = 0; r < 16; ++r) acc_sum += acc0[r] + acc1[r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc_sum;
}

double run_dbuf(const char* name, int blocks, int iters, float* out, const float* src, size_t src_floats) {
    const size_t lds = (size_t)2 * (128 * 36 + 32 * BN) * sizeof(float);
    hipFuncSetAttribute((const void*)k_dbuf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_dbuf, dim3(blocks), dim3(256), lds, 0, out, src, src_floats, iters);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_dbuf, dim3(blocks), dim3(256), lds, 0, out, src, src_floats, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = (double)blocks * iters * 4 * 32 * 64 * 32 * 2;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-46s blocks %4d  %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", name, blocks, ms, tf, tf / 157.3);
    return tf;
}

template <int V>
double run(const char* name, int blocks, int iters, float* out) {
    constexpr int ROWS = (V == 0 || V == 2) ? 128 : 256;
    constexpr int BK = (V == 0 || V == 2) ? 64 : 32;
    constexpr int LDA = BK + 4;
    const size_t lds = (size_t)(ROWS * LDA + BK * BN) * sizeof(float);
    hipFuncSetAttribute((const void*)k_loop<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_loop<V>, dim3(blocks), dim3(256), lds, 0, out, iters);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_loop<V>, dim3(blocks), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    // flops per block per iteration: wave tile rows x 64 cols x BK x 2, 4 waves
    const double wrows = (V == 0 || V == 2) ? 32 : 64;
    const double flops = (double)blocks * iters * 4 * wrows * 64 * BK * 2;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-34s blocks %4d  lds %6zu B  %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", name, blocks, lds, ms, tf, tf / 157.3);
    return tf;
}

int main(int argc, char** argv) {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 4096);
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    for (int pass = 0; pass < 1; ++pass) {
        run<0>("A 32x32x2  wave 32x64 (current)", 768, iters, out);
        run<1>("B 32x32x2  wave 64x64", 768, iters, out);
        run<2>("C 16x16x4  wave 32x64", 768, iters, out);
        run<3>("D 16x16x4  wave 64x64", 768, iters, out);
        run<0>("A 32x32x2  wave 32x64, 1 blk/CU", 256, iters * 3, out);
        run<1>("B 32x32x2  wave 64x64, 1 blk/CU", 256, iters * 3, out);
        run<3>("D 16x16x4  wave 64x64, 1 blk/CU", 256, iters * 3, out);
    }
    const size_t src_floats = (size_t)8 * 200 * 400 * 64;
    float* src;
    hipMalloc(&src, src_floats * sizeof(float));
    {
        std::vector<float> h(src_floats);
        unsigned x = 12345u;
        for (size_t i = 0; i < src_floats; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 16777216.f - 0.5f; }
        hipMemcpy(src, h.data(), src_floats * sizeof(float), hipMemcpyHostToDevice);
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int blocks : {768, 2500}) {
            const int it = blocks == 768 ? iters / 4 : 27;
            run_staged<0>("S0 loop A only", blocks, it, out, src, src_floats);
            run_staged<1>("S1 + 2 barriers per slab", blocks, it, out, src, src_floats);
            run_staged<2>("S2 + LDS stores", blocks, it, out, src, src_floats);
            run_staged<3>("S3 + global loads", blocks, it, out, src, src_floats);
            run_staged<4>("S4 + pipelined fragment reads (k_igemm)", blocks, it, out, src, src_floats);
            run_dbuf("S5 BK32 LDS double buffer, interleaved", blocks, 2 * it, out, src, src_floats);
            run_staged<5>("S6 = S4 with the A tile staged every 3rd slab", blocks, it, out, src, src_floats);
        }
    }
    hipFree(src);
    hipFree(out);
    return 0;
}
