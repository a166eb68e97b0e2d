// Sparse-exact VFE stack for gfx950 -- forward.
//
// Replaces, for one sample, the dense rank-6 graph of the reference
//   addVFELayer(6,32) -> addVFELayer(32,64) -> addFCN(64,64) -> MaxPoolingVFELayer(combine=True)
//   (model_training.py:155-186, :231-235; layers :32-61)
// which runs Dense(no bias)+BN+ReLU over 8*200*400*35 = 22.4 M rows.  Dense has no bias and there
// is no point mask, so dense rows fall in three classes that stay bit-identical through the stack:
// real point rows, ONE zero pad row per non-empty voxel (weight T - s_v) and ONE zero row for all
// empty voxels (weight T * n_empty, handled as a "virtual voxel" with ordinal V).  Only class
// representatives are evaluated; BN batch statistics weight them by multiplicity and divide by the
// dense row count (derivation + proof against dense autograd: oracle/vfe_sparse_ref.py).
//
// Mapping: one wave per voxel, lane = output channel, rows looped; the row's inputs are broadcast
// with v_readlane.  concat([repeat(max), pointwise]) @ W splits into a per-voxel constant
// (pooled half @ W[:C/2]) + a per-row half, halving the FMAs.  relu(BN(.)) is monotone per channel,
// so max_t relu(BN(y_t)) == relu(BN(max_t y_t or min_t y_t)): only per-voxel max/min of the pre-BN
// value are kept ("ymm"), nothing per row is stored and each stage recomputes the cheap lower layers.
//
// Training needs a grid-wide BN reduction per layer => 3 stage kernels + 3 finalize kernels;
// inference (moving statistics) runs the three passes inside one kernel.  The last kernel streams
// the dense (D,H,W,64) grid: 164 MB for the Lyft grid, the HBM-bound part of this file (empty cells
// hold the non-zero constant relu(BN3(.)), so the write is compulsory).
#include "vfe_common.h"

namespace lisec {
namespace {

// STAGE 1/2/3: training stage kernels (batch statistics of layer STAGE are reduced);
// STAGE 0: inference, all three passes, statistics come from bnstate (moving averages).
template <int STAGE>
__global__ void __launch_bounds__(kVfeThreads)
k_vfe_stage(VfeIn in, const float* __restrict__ W1, const float* __restrict__ W2,
            const float* __restrict__ W3, const float* __restrict__ bn1, const float* __restrict__ bn2,
            float* __restrict__ ymm1, float* __restrict__ ymm2, float* __restrict__ ymm3,
            double* __restrict__ partials) {
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c1 = lane & 15, c2 = lane & 31;
    VfeWeights W;
    W.load(W1, W2, W3, STAGE);
    float sc1 = 0, sh1 = 0, sc2 = 0, sh2 = 0;
    if (STAGE != 1) { sc1 = bn1[c1]; sh1 = bn1[16 + c1]; }
    if (STAGE == 0 || STAGE == 3) { sc2 = bn2[c2]; sh2 = bn2[32 + c2]; }
    // the pad row after layer 1 is the same everywhere: relu(BN1(0)) = relu(shift1)
    const float a1pad = fmaxf(sh1, 0.0f);
    float A2pad = 0.0f;                                     // a1pad @ W2[16:, :]
    if (STAGE != 1) {
#pragma unroll
        for (int k = 0; k < 16; ++k) A2pad = fmaf(rl(a1pad, k), W.w2a[k], A2pad);
    }
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nE = in.ncells - V;
    const int nvox = V + (nE > 0 ? 1 : 0);
    const int nwaves = gridDim.x * (kVfeThreads / 64);
    double s1 = 0.0, s2 = 0.0;
    for (int v = blockIdx.x * (kVfeThreads / 64) + w; v < nvox; v += nwaves) {
        const bool virt = v == V;
        const int s = virt ? 0 : in.npts[v];
        const int rs = virt ? 0 : in.row_start[v];
        const bool has_pad = virt || s < in.T;
        const double wpad = virt ? (double)in.T * (double)nE : (double)(in.T - s);
        float xr[6] = {0, 0, 0, 0, 0, 0};
        if (lane < s) {
#pragma unroll
            for (int k = 0; k < 6; ++k) xr[k] = in.rows[(size_t)(rs + lane) * 6 + k];
        }
        // ---- pass 1: y1 = x @ W1 --------------------------------------------------------------
        float mx1, mn1;
        if (STAGE == 0 || STAGE == 1) {
            mx1 = has_pad ? 0.0f : -INFINITY;               // pad row: 0 @ W1 == 0
            mn1 = has_pad ? 0.0f : INFINITY;
            for (int t = 0; t < s; ++t) {
                float y = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) y = fmaf(rl(xr[k], t), W.w1[k], y);
                mx1 = fmaxf(mx1, y); mn1 = fminf(mn1, y);
                if (STAGE == 1) { s1 += (double)y; s2 += (double)y * (double)y; }
            }
            if (STAGE == 1) {
                if (lane < 16) { ymm1[(size_t)v * 32 + lane] = mx1; ymm1[(size_t)v * 32 + 16 + lane] = mn1; }
                continue;
            }
        } else {
            mx1 = ymm1[(size_t)v * 32 + c1]; mn1 = ymm1[(size_t)v * 32 + 16 + c1];
        }
        // ---- pass 2: y2 = [pool1, a1] @ W2 -----------------------------------------------------
        const float pool1 = pool_from(mx1, mn1, sc1, sh1);
        float P2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) P2 = fmaf(rl(pool1, k), W.w2p[k], P2);
        float mx2, mn2;
        if (STAGE == 0 || STAGE == 2) {
            const float y2pad = P2 + A2pad;
            mx2 = has_pad ? y2pad : -INFINITY;
            mn2 = has_pad ? y2pad : INFINITY;
            if (STAGE == 2 && has_pad) { s1 += wpad * (double)y2pad; s2 += wpad * (double)y2pad * (double)y2pad; }
            for (int t = 0; t < s; ++t) {
                float y1 = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) y1 = fmaf(rl(xr[k], t), W.w1[k], y1);
                const float a1 = bnrelu(y1, sc1, sh1);
                float y = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; ++k) y = fmaf(rl(a1, k), W.w2a[k], y);
                y += P2;
                mx2 = fmaxf(mx2, y); mn2 = fminf(mn2, y);
                if (STAGE == 2) { s1 += (double)y; s2 += (double)y * (double)y; }
            }
            if (STAGE == 2) {
                if (lane < 32) { ymm2[(size_t)v * 64 + lane] = mx2; ymm2[(size_t)v * 64 + 32 + lane] = mn2; }
                continue;
            }
        } else {
            mx2 = ymm2[(size_t)v * 64 + c2]; mn2 = ymm2[(size_t)v * 64 + 32 + c2];
        }
        // ---- pass 3: y3 = [pool2, a2] @ W3 -----------------------------------------------------
        const float pool2 = pool_from(mx2, mn2, sc2, sh2);
        float P3 = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) P3 = fmaf(rl(pool2, k), W.w3p[k], P3);
        float mx3 = -INFINITY, mn3 = INFINITY;
        if (has_pad) {
            const float a2pad = bnrelu(P2 + A2pad, sc2, sh2);
            float y = 0.0f;
#pragma unroll
            for (int k = 0; k < 32; ++k) y = fmaf(rl(a2pad, k), W.w3a[k], y);
            y += P3;
            mx3 = y; mn3 = y;
            if (STAGE == 3) { s1 += wpad * (double)y; s2 += wpad * (double)y * (double)y; }
        }
        for (int t = 0; t < s; ++t) {
            float y1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) y1 = fmaf(rl(xr[k], t), W.w1[k], y1);
            const float a1 = bnrelu(y1, sc1, sh1);
            float y2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) y2 = fmaf(rl(a1, k), W.w2a[k], y2);
            y2 += P2;
            const float a2 = bnrelu(y2, sc2, sh2);
            float y = 0.0f;
#pragma unroll
            for (int k = 0; k < 32; ++k) y = fmaf(rl(a2, k), W.w3a[k], y);
            y += P3;
            mx3 = fmaxf(mx3, y); mn3 = fminf(mn3, y);
            if (STAGE == 3) { s1 += (double)y; s2 += (double)y * (double)y; }
        }
        ymm3[(size_t)v * 128 + lane] = mx3;
        ymm3[(size_t)v * 128 + 64 + lane] = mn3;
    }
    if (STAGE != 0) {
        constexpr int C = STAGE == 1 ? 16 : (STAGE == 2 ? 32 : 64);
        __shared__ double red[2][kVfeThreads / 64][64];
        red[0][w][lane] = s1; red[1][w][lane] = s2;
        __syncthreads();
        if (threadIdx.x < 2 * C) {
            int q = threadIdx.x / C, c = threadIdx.x % C;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < kVfeThreads / 64; ++k) a += red[q][k][c];
            partials[((size_t)blockIdx.x * 2 + q) * C + c] = a;
        }
    }
}

// Dense (ncells, 64) grid: occupied cells from their voxel's ymm3, empty cells from the virtual voxel.
__global__ void __launch_bounds__(256)
k_vfe_grid(const int* __restrict__ info, const int* __restrict__ cell_voxel, int ncells, int cap,
           const float* __restrict__ ymm3, const float* __restrict__ bn3, float* __restrict__ grid) {
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const int q = threadIdx.x & 15;
    const float4 sc = reinterpret_cast<const float4*>(bn3)[q];
    const float4 sh = reinterpret_cast<const float4*>(bn3 + 64)[q];
    const long long total = (long long)ncells * 16;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cell = (int)(i >> 4);
        int v = cell_voxel[cell];
        if (v < 0) v = V;
        const float4 mx = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128)[q];
        const float4 mn = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128 + 64)[q];
        float4 o;
        o.x = pool_from(mx.x, mn.x, sc.x, sh.x);
        o.y = pool_from(mx.y, mn.y, sc.y, sh.y);
        o.z = pool_from(mx.z, mn.z, sc.z, sh.z);
        o.w = pool_from(mx.w, mn.w, sc.w, sh.w);
        reinterpret_cast<float4*>(grid)[i] = o;
    }
}

// compact per-voxel outputs: vout[v] = the grid value of voxel v (v == V: the empty-cell constant),
// delta[v] = vout[v] - vout[V]  (what the first Conv3D's sparse backward contracts against)
__global__ void __launch_bounds__(256)
k_vfe_vout(const int* __restrict__ info, int cap, const float* __restrict__ ymm3, const float* __restrict__ bn3,
           float* __restrict__ vout, float* __restrict__ delta) {
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const int q = threadIdx.x & 15;
    const float4 sc = reinterpret_cast<const float4*>(bn3)[q];
    const float4 sh = reinterpret_cast<const float4*>(bn3 + 64)[q];
    auto value = [&](int v) {
        const float4 mx = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128)[q];
        const float4 mn = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128 + 64)[q];
        return make_float4(pool_from(mx.x, mn.x, sc.x, sh.x), pool_from(mx.y, mn.y, sc.y, sh.y),
                           pool_from(mx.z, mn.z, sc.z, sh.z), pool_from(mx.w, mn.w, sc.w, sh.w));
    };
    const float4 c = value(V);
    const long long total = (long long)(V + 1) * 16;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int v = (int)(i >> 4);
        const float4 o = value(v);
        reinterpret_cast<float4*>(vout)[i] = o;
        reinterpret_cast<float4*>(delta)[i] = make_float4(o.x - c.x, o.y - c.y, o.z - c.z, o.w - c.w);
    }
}

}  // namespace

}  // namespace lisec

using namespace lisec;

extern "C" size_t lisec_vfe_saved_field_offset(int cap_voxels, int field) {
    VfeSaved sv(nullptr, cap_voxels < 0 ? 0 : cap_voxels);
    const float* base = nullptr;
    switch (field) {
        case LISEC_VFE_SAVED_VOUT: return (size_t)(sv.vout - base);
        case LISEC_VFE_SAVED_DELTA: return (size_t)(sv.delta - base);
        default: return (size_t)-1;
    }
}

extern "C" size_t lisec_vfe_saved_floats(int cap_voxels) {
    if (cap_voxels < 0) return 0;
    return VfeSaved(nullptr, cap_voxels).floats;
}

extern "C" size_t lisec_vfe_workspace_bytes(void) {
    return align_up(sizeof(double) * (size_t)kVfeBlocks * 2 * 64, 256);
}

extern "C" int lisec_vfe_forward(const lisec_vfe_params* p, const int32_t* info,
                                 const int32_t* cell_voxel, const int32_t* npts,
                                 const int32_t* row_start, const float* rows, int ncells, int T,
                                 int cap_voxels, int training, float* saved, void* workspace,
                                 size_t workspace_bytes, float* grid, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(p && info && cell_voxel && npts && row_start && rows && saved && workspace && grid,
                    "NULL pointer");
    LISEC_CHECK_ARG(ncells > 0 && T >= 1 && T <= 64 && cap_voxels >= 0, "bad sizes");
    for (int i = 0; i < 3; ++i)
        LISEC_CHECK_ARG(p->kernel[i] && p->gamma[i] && p->beta[i] && p->moving_mean[i] && p->moving_var[i],
                        "NULL VFE parameter pointer");
    if (workspace_bytes < lisec_vfe_workspace_bytes()) {
        set_error("vfe workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    VfeSaved sv(saved, cap_voxels);
    VfeIn in{info, npts, row_start, rows, ncells, T, cap_voxels};
    double* parts = static_cast<double*>(workspace);
    const double N = (double)ncells * (double)T;        // dense rows Keras reduces over (B = 1)
    dim3 g(kVfeBlocks), b(kVfeThreads);
    if (training) {
        hipLaunchKernelGGL(k_vfe_stage<1>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2],
                           sv.bn1, sv.bn2, sv.ymm1, sv.ymm2, sv.ymm3, parts);
        LISEC_LAUNCH_CHECK();
        if (int rc = launch_bn_finalize(parts, kVfeBlocks, 16, N, p->gamma[0], p->beta[0], p->moving_mean[0],
                                        p->moving_var[0], /*unbiased=*/0, sv.bn1, st)) return rc;
        hipLaunchKernelGGL(k_vfe_stage<2>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2],
                           sv.bn1, sv.bn2, sv.ymm1, sv.ymm2, sv.ymm3, parts);
        LISEC_LAUNCH_CHECK();
        if (int rc = launch_bn_finalize(parts, kVfeBlocks, 32, N, p->gamma[1], p->beta[1], p->moving_mean[1],
                                        p->moving_var[1], 0, sv.bn2, st)) return rc;
        hipLaunchKernelGGL(k_vfe_stage<3>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2],
                           sv.bn1, sv.bn2, sv.ymm1, sv.ymm2, sv.ymm3, parts);
        LISEC_LAUNCH_CHECK();
        if (int rc = launch_bn_finalize(parts, kVfeBlocks, 64, N, p->gamma[2], p->beta[2], p->moving_mean[2],
                                        p->moving_var[2], 0, sv.bn3, st)) return rc;
    } else {
        float* bns[3] = {sv.bn1, sv.bn2, sv.bn3};
        const int C[3] = {16, 32, 64};
        for (int i = 0; i < 3; ++i)
            if (int rc = launch_bn_fold(p->gamma[i], p->beta[i], p->moving_mean[i], p->moving_var[i], C[i],
                                        bns[i], st)) return rc;
        hipLaunchKernelGGL(k_vfe_stage<0>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2],
                           sv.bn1, sv.bn2, sv.ymm1, sv.ymm2, sv.ymm3, parts);
        LISEC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_vfe_grid, dim3(4096), dim3(256), 0, st, info, cell_voxel, ncells, cap_voxels,
                       sv.ymm3, sv.bn3, grid);
    {
        int vb = cdiv((long long)(cap_voxels + 1) * 16, 256);
        if (vb > 1024) vb = 1024;
        hipLaunchKernelGGL(k_vfe_vout, dim3(vb), dim3(256), 0, st, info, cap_voxels, sv.ymm3, sv.bn3, sv.vout, sv.delta);
    }
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_vfe_grid_from_saved(const int32_t* info, const int32_t* cell_voxel, int ncells,
                                         int cap_voxels, const float* saved, float* grid, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(info && cell_voxel && saved && grid && ncells > 0 && cap_voxels >= 0, "bad arguments");
    VfeSaved sv(const_cast<float*>(saved), cap_voxels);
    hipLaunchKernelGGL(k_vfe_grid, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream_), info, cell_voxel,
                       ncells, cap_voxels, sv.ymm3, sv.bn3, grid);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
