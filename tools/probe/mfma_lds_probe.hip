// What does a lone wave per SIMD pay per K step of the implicit-GEMM loop besides its 64 MFMAs?  Variants of one loop:
//   A  64 MFMAs on two accumulators, operands in registers                      (the issue rate)
//   B  + the step's 24 ds_read_b128 fragment reads, software-pipelined as in igemm.hip, operands fed from them
//   C  B + one __syncthreads per step
//   D  C + four more waves per workgroup that stream 48 KB per step from global memory into the other LDS image
// (measurement aid, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDA = 68, A_FLOATS = 128 * LDA, B_FLOATS = 64 * 64, IMG = A_FLOATS + B_FLOATS;

template <int V>
__global__ void __launch_bounds__(V >= 3 ? 512 : 256) k_step(const float* __restrict__ src, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * IMG; i += blockDim.x) smem[i] = 1e-3f * (i & 255);
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    const bool loader = V >= 3 && wave >= 4;
    const int stid = loader ? tid - 256 : tid;
    const int aoff = ((wave & 3) * 32 + (lane & 31)) * LDA + 4 * (lane >> 5), boff = ((lane >> 5) * 64 + (lane & 31)) * 4;
    int img = 0;
    float4 ra[8], rb[4];
    const float* gsrc = src + (size_t)blockIdx.x * 65536 + stid * 4;
    if (loader && V != 7) {
#pragma unroll
        for (int p = 0; p < 8; ++p) ra[p] = *reinterpret_cast<const float4*>(gsrc + p * 1024);
#pragma unroll
        for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const float4*>(gsrc + 8192 + p * 1024);
    }
    for (int s = 0; s < steps; ++s) {
        const float* aRow = smem + img * IMG + aoff;
        const float* bCol = smem + img * IMG + A_FLOATS + boff;
        if (loader && V == 7) {
            // H: the same 48 KB (L2 hits) straight into LDS (global_load_lds_dwordx4: lane i's 16 bytes land at base + 16 i),
            // no VGPR round trip, no ds_write
            typedef __attribute__((address_space(1))) const void* gptr;
            typedef __attribute__((address_space(3))) void* lptr;
            float* dst = smem + (img ^ 1) * IMG + (wave - 4) * 12 * 256;          // 12 KB per loader wave, wave-uniform base
#pragma unroll
            for (int p = 0; p < 12; ++p)
                __builtin_amdgcn_global_load_lds((gptr)(gsrc + p * 1024), (lptr)(dst + p * 256), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            img ^= 1;
            __syncthreads();
            continue;
        }
        if (loader) {
            float* sA = smem + (img ^ 1) * IMG; float* sB = sA + A_FLOATS;
            if (V != 6) {                           // (G: the loaders only take part in the barrier)
#pragma unroll
                for (int p = 0; p < 8; ++p) *reinterpret_cast<float4*>(sA + (p * 16 + (stid >> 4)) * LDA + (stid & 15) * 4) = ra[p];
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(sB + ((stid >> 6) * 64 + (stid & 63)) * 4 + p * 1024) = rb[p];
            }
            if (V != 4 && V != 6) {                           // (E: LDS stores only, the registers are never reloaded)
                const float* gs = gsrc + (V == 5 ? 0 : ((s + 1) & 3) * 12288);      // (F: the same 48 KB every step: L2 hits)
#pragma unroll
                for (int p = 0; p < 8; ++p) ra[p] = *reinterpret_cast<const float4*>(gs + p * 1024);
#pragma unroll
                for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const float4*>(gs + 8192 + p * 1024);
            }
            img ^= 1;
            __syncthreads();
            continue;
        }
        if (V == 0) {
            const float a = lane * 1e-3f, b = lane * 2e-3f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
            }
        } else {
            float4 a = *reinterpret_cast<const float4*>(aRow);
            float4 b0 = *reinterpret_cast<const float4*>(bCol);
            float4 b1 = *reinterpret_cast<const float4*>(bCol + 32 * 4);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                float4 an = a, b0n = b0, b1n = b1;
                if (kc + 1 < 8) {
                    an = *reinterpret_cast<const float4*>(aRow + (kc + 1) * 8);
                    b0n = *reinterpret_cast<const float4*>(bCol + (kc + 1) * 2 * 64 * 4);
                    b1n = *reinterpret_cast<const float4*>(bCol + (kc + 1) * 2 * 64 * 4 + 32 * 4);
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
        }
        if (V >= 2) { img ^= 1; __syncthreads(); }
    }
    if (loader) return;
    float sum = 0;
    for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int V>
void run(const char* what, const float* src, float* out) {
    const int steps = 400, grid = 256;
    const size_t lds = 2 * IMG * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k_step<V><<<grid, V >= 3 ? 512 : 256, lds>>>(src, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-78s %.3f us per 64-MFMA step (%.1f ns per MFMA)\n", what, ms * 1e3 / steps, ms * 1e6 / steps / 64);
}

int main() {
    float *src, *out;
    hipMalloc(&src, (size_t)256 * 65536 * 4); hipMalloc(&out, 256 * 256 * 4);
    hipMemset(src, 0, (size_t)256 * 65536 * 4);
    run<0>("A  64 MFMAs, operands in registers", src, out);
    run<1>("B  + 24 ds_read_b128 fragment reads per step, pipelined one chunk ahead", src, out);
    run<2>("C  B + one barrier per step", src, out);
    run<3>("D  C + four loader waves streaming 48 KB per step (HBM) into the other LDS image", src, out);
    run<4>("E  C + four loader waves that only write 48 KB per step into the other LDS image", src, out);
    run<5>("F  C + four loader waves re-reading the same 48 KB (L2 hits) into the other LDS image", src, out);
    run<6>("G  C + four more waves that only join the barrier", src, out);
    run<7>("H  C + four loader waves: the same 48 KB (L2 hits) by global_load_lds_dwordx4, waited for in the step", src, out);
    return 0;
}
