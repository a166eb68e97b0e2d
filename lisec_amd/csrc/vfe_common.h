// Internal: device helpers shared by the VFE forward (vfe.hip) and backward (vfe_bwd.hip) kernels.
#pragma once
#include "bn.h"

namespace lisec {

constexpr int kVfeBlocks = 512;
constexpr int kVfeThreads = 256;     // 4 waves

__device__ __forceinline__ float rl(float v, int k) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
__device__ __forceinline__ float bnrelu(float y, float sc, float sh) { return fmaxf(fmaf(y, sc, sh), 0.0f); }
__device__ __forceinline__ float pool_from(float ymax, float ymin, float sc, float sh) {
    return bnrelu(sc >= 0.0f ? ymax : ymin, sc, sh);
}

struct VfeIn {
    const int* info; const int* npts; const int* row_start; const float* rows;
    int ncells, T, cap;
};

struct VfeWeights {          // per lane: column (lane & (C-1)) of each Dense kernel
    float w1[6];
    float w2p[16], w2a[16];
    float w3p[32], w3a[32];
    // pooled: also the halves that multiply the pooled (per-voxel) inputs; the forward keeps those in LDS instead
    __device__ void load(const float* W1, const float* W2, const float* W3, int stage, bool pooled = true) {
        const int lane = lane_id();
#pragma unroll
        for (int k = 0; k < 6; ++k) w1[k] = W1[k * 16 + (lane & 15)];
        if (stage != 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (pooled) w2p[k] = W2[k * 32 + (lane & 31)];
                w2a[k] = W2[(16 + k) * 32 + (lane & 31)];
            }
        }
        if (stage == 0 || stage == 3) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                if (pooled) w3p[k] = W3[k * 64 + lane];
                w3a[k] = W3[(32 + k) * 64 + lane];
            }
        }
    }
};


struct VfeSaved {            // layout of the caller-owned `saved` float buffer (lisec_vfe_saved_floats / _rows)
    float *bn1, *bn2, *bn3, *ymm1, *ymm2, *ymm3;
    float *vout, *delta;     // per-voxel grid value (row V = the empty-cell constant) and vout[v] - vout[V]
    // per-row extras of a training forward (only when the buffer was sized with the row capacity, n_points > 0):
    // what the tiled (MFMA) backward reads instead of recomputing the layers and searching for the max-pool winners.
    //   arg1/2/3  uint8[(cap+1)][2][C]: class-row slot (0 = the pad row, t+1 = real row t) of the FIRST row holding the
    //             per-voxel maximum ([0]) / minimum ([1]) of the pre-BN value of layer 1/2/3
    //   y2rows    float[slots][32]: pre-BN output of layer 2 of every class row; slot of row j of voxel v =
    //             row_start[v] + v + j (slot 0 of a voxel is its pad row, unused when the voxel is full)
    unsigned char *arg1, *arg2, *arg3;
    float* y2rows;
    size_t floats;
    static size_t slots(int cap, int n_points) { return (size_t)n_points + (size_t)cap + 2; }
    VfeSaved(float* base, int cap, int n_points = 0) {
        size_t o = 0;
        bn1 = base + o; o += 4 * 16;
        bn2 = base + o; o += 4 * 32;
        bn3 = base + o; o += 4 * 64;
        ymm1 = base + o; o += (size_t)(cap + 1) * 32;
        ymm2 = base + o; o += (size_t)(cap + 1) * 64;
        ymm3 = base + o; o += (size_t)(cap + 1) * 128;
        vout = base + o; o += (size_t)(cap + 1) * 64;
        delta = base + o; o += (size_t)(cap + 1) * 64;
        arg1 = arg2 = arg3 = nullptr; y2rows = nullptr;
        if (n_points > 0) {
            arg1 = reinterpret_cast<unsigned char*>(base + o); o += (size_t)(cap + 1) * 32 / 4;
            arg2 = reinterpret_cast<unsigned char*>(base + o); o += (size_t)(cap + 1) * 64 / 4;
            arg3 = reinterpret_cast<unsigned char*>(base + o); o += (size_t)(cap + 1) * 128 / 4;
            y2rows = base + o; o += slots(cap, n_points) * 32;
        }
        floats = o;
    }
};

}  // namespace lisec
