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
// Training needs a grid-wide BN reduction per layer => 3 stage kernels; the statistics of layer L are
// finalised in the PROLOGUE of stage L+1 (every workgroup sums the 256 per-workgroup partials in the
// same fixed order while its weight loads are in flight; workgroup 0 also stores bnstate and updates
// the moving statistics), so only layer 3 needs a finaliser launch of its own (one workgroup per
// channel) in front of the grid writer: 5 launches.  Inference (moving statistics) runs the three
// passes inside one kernel: 2 launches.  Voxel metadata and input rows are prefetched one and two
// voxels ahead (the stage kernels are latency-, not throughput-bound on a 20 k-point sweep).  The
// last kernel streams the dense (D,H,W,64) grid: 164 MB for the Lyft grid, the HBM-bound part of this
// file (empty cells hold the non-zero constant relu(BN3(.)), so the write is compulsory), and writes
// the compact per-voxel outputs (vout / delta) on the side.
#include "vfe_common.h"

namespace lisec {
namespace {

constexpr int kFwdBlocks = 256;      // one workgroup per CU; also the number of statistic partials per layer
constexpr int kFwdThreads = 512;     // 8 waves
constexpr int kFwdWaves = kFwdThreads / 64;

// Sums parts[nparts][2][C] (double) over the workgroups in a FIXED order (thread (grp, col) takes parts grp,
// grp + ngrp, ... in index order, then the groups are combined in index order): every workgroup of the
// consuming kernel computes bit-identical scale / shift.  Result in LDS: sbn[0..C) = scale, sbn[C..2C) = shift.
// Workgroup 0 also stores the bnstate the backward reads and updates the moving statistics (biased batch
// variance: the rank-6 VFE path of Keras is not the fused one).
template <int C>
__device__ __forceinline__ void block_finalize(const double* __restrict__ parts, int nparts, double N,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float* __restrict__ mmean, float* __restrict__ mvar,
                                               float* __restrict__ bnsaved, float* sbn, double* red) {
    constexpr int cols = 2 * C, ngrp = kFwdThreads / cols;
    const int col = threadIdx.x % cols, grp = threadIdx.x / cols;
    double a = 0.0;
    int b = grp;
    for (; b + 3 * ngrp < nparts; b += 4 * ngrp) {
        const double v0 = parts[(size_t)b * cols + col];
        const double v1 = parts[(size_t)(b + ngrp) * cols + col];
        const double v2 = parts[(size_t)(b + 2 * ngrp) * cols + col];
        const double v3 = parts[(size_t)(b + 3 * ngrp) * cols + col];
        a += v0; a += v1; a += v2; a += v3;
    }
    for (; b < nparts; b += ngrp) a += parts[(size_t)b * cols + col];
    red[grp * cols + col] = a;
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int g = 0; g < ngrp; ++g) { s1 += red[g * cols + c]; s2 += red[g * cols + C + c]; }
        const double mean = s1 / N;
        double var = s2 / N - mean * mean;             // biased; fp64 so the cancellation is harmless
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + (double)kBnEps);
        const double scale = (double)gamma[c] * inv;
        const float fsc = (float)scale, fsh = (float)((double)beta[c] - mean * scale);
        sbn[c] = fsc;
        sbn[C + c] = fsh;
        if (blockIdx.x == 0) {
            bnsaved[c] = fsc; bnsaved[C + c] = fsh; bnsaved[2 * C + c] = (float)mean; bnsaved[3 * C + c] = (float)inv;
            mmean[c] = (float)((double)mmean[c] * (double)kBnMomentum + mean * (1.0 - (double)kBnMomentum));
            mvar[c] = (float)((double)mvar[c] * (double)kBnMomentum + var * (1.0 - (double)kBnMomentum));
        }
    }
    __syncthreads();
}

// inference: scale / shift from the moving statistics (what lisec_bn_fold computes), into LDS (+ bnsaved by workgroup 0)
template <int C>
__device__ __forceinline__ void block_fold(const float* __restrict__ gamma, const float* __restrict__ beta,
                                           const float* __restrict__ mmean, const float* __restrict__ mvar,
                                           float* __restrict__ bnsaved, float* sbn) {
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        const double inv = 1.0 / sqrt((double)mvar[c] + (double)kBnEps);
        const double scale = (double)gamma[c] * inv;
        const float fsc = (float)scale, fsh = (float)((double)beta[c] - (double)mmean[c] * scale);
        sbn[c] = fsc;
        sbn[C + c] = fsh;
        if (blockIdx.x == 0) {
            bnsaved[c] = fsc; bnsaved[C + c] = fsh; bnsaved[2 * C + c] = mmean[c]; bnsaved[3 * C + c] = (float)inv;
        }
    }
}

struct StageBn {            // BatchNormalization variables of the three layers + where their state is saved
    const float* gamma[3];
    const float* beta[3];
    float* mmean[3];
    float* mvar[3];
    float* saved[3];        // sv.bn1, sv.bn2, sv.bn3
};

// STAGE 1/2/3: training stage kernels (batch statistics of layer STAGE are reduced into parts_out; the
// statistics of layer STAGE-1 are finalised from parts_in in the prologue);
// STAGE 0: inference, all three passes, statistics from the moving averages.
template <int STAGE>
__global__ void __launch_bounds__(kFwdThreads)
k_vfe_stage(VfeIn in, const float* __restrict__ W1, const float* __restrict__ W2,
            const float* __restrict__ W3, StageBn bn, double N,
            float* __restrict__ ymm1, float* __restrict__ ymm2, float* __restrict__ ymm3,
            const double* __restrict__ parts_in, double* __restrict__ parts_out) {
    __shared__ float sbn1[32], sbn2[64];
    __shared__ double red[kFwdThreads > 2 * kFwdWaves * 64 ? kFwdThreads : 2 * kFwdWaves * 64];
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c1 = lane & 15, c2 = lane & 31;
    VfeWeights W;
    W.load(W1, W2, W3, STAGE);                        // issued first: in flight while the statistics are summed
    if (STAGE == 0) {
        block_fold<16>(bn.gamma[0], bn.beta[0], bn.mmean[0], bn.mvar[0], bn.saved[0], sbn1);
        block_fold<32>(bn.gamma[1], bn.beta[1], bn.mmean[1], bn.mvar[1], bn.saved[1], sbn2);
        __syncthreads();
    } else if (STAGE == 2) {
        block_finalize<16>(parts_in, kFwdBlocks, N, bn.gamma[0], bn.beta[0], bn.mmean[0], bn.mvar[0], bn.saved[0],
                           sbn1, red);
    } else if (STAGE == 3) {
        if (threadIdx.x < 32) sbn1[threadIdx.x] = bn.saved[0][threadIdx.x];      // written by stage 2's workgroup 0
        block_finalize<32>(parts_in, kFwdBlocks, N, bn.gamma[1], bn.beta[1], bn.mmean[1], bn.mvar[1], bn.saved[1],
                           sbn2, red);
    }
    float sc1 = 0, sh1 = 0, sc2 = 0, sh2 = 0;
    if (STAGE != 1) { sc1 = sbn1[c1]; sh1 = sbn1[16 + c1]; }
    if (STAGE == 0 || STAGE == 3) { sc2 = sbn2[c2]; sh2 = sbn2[32 + c2]; }
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
    const int nwaves = gridDim.x * kFwdWaves;
    double s1 = 0.0, s2 = 0.0;

    // software pipeline over this wave's voxels: metadata two voxels ahead, input rows (and the saved
    // per-voxel max/min of the lower layers) one voxel ahead
    auto load_meta = [&](int v, int& s, int& rs) {
        s = 0; rs = 0;
        if (v < V) { s = in.npts[v]; rs = in.row_start[v]; }
    };
    auto load_rows = [&](int s, int rs, float (&x)[6]) {
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = 0.0f;
        if (lane < s) {
            const float2* r = reinterpret_cast<const float2*>(in.rows + (size_t)(rs + lane) * 6);   // 24-byte rows
            const float2 p0 = r[0], p1 = r[1], p2 = r[2];
            x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y;
        }
    };
    auto load_ymm = [&](int v, float& mx1, float& mn1, float& mx2, float& mn2) {
        mx1 = mn1 = mx2 = mn2 = 0.0f;
        if (v < nvox) {
            if (STAGE == 2 || STAGE == 3) { mx1 = ymm1[(size_t)v * 32 + c1]; mn1 = ymm1[(size_t)v * 32 + 16 + c1]; }
            if (STAGE == 3) { mx2 = ymm2[(size_t)v * 64 + c2]; mn2 = ymm2[(size_t)v * 64 + 32 + c2]; }
        }
    };
    int v = blockIdx.x * kFwdWaves + w;
    int s_cur, rs_cur, s_nxt, rs_nxt;
    load_meta(v, s_cur, rs_cur);
    load_meta(v + nwaves, s_nxt, rs_nxt);
    float xr[6], pmx1, pmn1, pmx2, pmn2;
    load_rows(s_cur, rs_cur, xr);
    load_ymm(v, pmx1, pmn1, pmx2, pmn2);
    for (; v < nvox; v += nwaves) {
        int s_n2, rs_n2;
        load_meta(v + 2 * nwaves, s_n2, rs_n2);
        float xn[6], nmx1, nmn1, nmx2, nmn2;
        load_rows(s_nxt, rs_nxt, xn);
        load_ymm(v + nwaves, nmx1, nmn1, nmx2, nmn2);

        const bool virt = v == V;
        const int s = s_cur;
        const bool has_pad = virt || s < in.T;
        const double wpad = virt ? (double)in.T * (double)nE : (double)(in.T - s);
        float mx1 = pmx1, mn1 = pmn1, mx2 = pmx2, mn2 = pmn2;
        bool done = false;
        // ---- pass 1: y1 = x @ W1 --------------------------------------------------------------
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
                done = true;
            }
        }
        if (!done) {
            // ---- pass 2: y2 = [pool1, a1] @ W2 -------------------------------------------------
            const float pool1 = pool_from(mx1, mn1, sc1, sh1);
            float P2 = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) P2 = fmaf(rl(pool1, k), W.w2p[k], P2);
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
                    done = true;
                }
            }
            if (!done) {
                // ---- pass 3: y3 = [pool2, a2] @ W3 ---------------------------------------------
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
        }
        s_cur = s_nxt; rs_cur = rs_nxt; s_nxt = s_n2; rs_nxt = rs_n2;
#pragma unroll
        for (int k = 0; k < 6; ++k) xr[k] = xn[k];
        pmx1 = nmx1; pmn1 = nmn1; pmx2 = nmx2; pmn2 = nmn2;
    }
    if (STAGE != 0) {
        constexpr int C = STAGE == 1 ? 16 : (STAGE == 2 ? 32 : 64);
        __syncthreads();                                     // `red` may still be read by a slow wave's prologue
        double* r0 = red;                                    // [2][kFwdWaves][64]
        r0[(0 * kFwdWaves + w) * 64 + lane] = s1;
        r0[(1 * kFwdWaves + w) * 64 + lane] = s2;
        __syncthreads();
        if (threadIdx.x < 2 * C) {
            const int q = threadIdx.x / C, c = threadIdx.x % C;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < kFwdWaves; ++k) a += r0[(q * kFwdWaves + k) * 64 + c];
            parts_out[((size_t)blockIdx.x * 2 + q) * C + c] = a;
        }
    } else if (blockIdx.x == 0) {
        // inference: the grid writer reads bn3 from `saved`
        __shared__ float sbn3[128];
        block_fold<64>(bn.gamma[2], bn.beta[2], bn.mmean[2], bn.mvar[2], bn.saved[2], sbn3);
    }
}

// Statistics of the last layer: one workgroup per channel sums the kFwdBlocks partials (fixed tree order).
__global__ void __launch_bounds__(kFwdBlocks)
k_vfe_final3(const double* __restrict__ parts, double N, const float* __restrict__ gamma,
             const float* __restrict__ beta, float* __restrict__ mmean, float* __restrict__ mvar,
             float* __restrict__ st) {
    constexpr int C = 64;
    __shared__ double r1[kFwdBlocks], r2[kFwdBlocks];
    const int c = blockIdx.x, t = threadIdx.x;
    r1[t] = parts[((size_t)t * 2 + 0) * C + c];
    r2[t] = parts[((size_t)t * 2 + 1) * C + c];
    __syncthreads();
    for (int o = kFwdBlocks / 2; o > 0; o >>= 1) {
        if (t < o) { r1[t] += r1[t + o]; r2[t] += r2[t + o]; }
        __syncthreads();
    }
    if (t == 0) {
        const double mean = r1[0] / N;
        double var = r2[0] / N - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = 1.0 / sqrt(var + (double)kBnEps);
        const double scale = (double)gamma[c] * inv;
        st[c] = (float)scale;
        st[C + c] = (float)((double)beta[c] - mean * scale);
        st[2 * C + c] = (float)mean;
        st[3 * C + c] = (float)inv;
        mmean[c] = (float)((double)mmean[c] * (double)kBnMomentum + mean * (1.0 - (double)kBnMomentum));
        mvar[c] = (float)((double)mvar[c] * (double)kBnMomentum + var * (1.0 - (double)kBnMomentum));
    }
}

// Dense (ncells, 64) grid: occupied cells from their voxel's ymm3, empty cells from the virtual voxel.
// On the side (vout != NULL) the compact per-voxel outputs: vout[v] = the grid value of voxel v (v == V: the
// constant every empty cell holds, 0 when the grid has no empty cell), delta[v] = vout[v] - vout[V] (what the
// first Conv3D's sparse backward contracts against).
__global__ void __launch_bounds__(256)
k_vfe_grid(const int* __restrict__ info, const int* __restrict__ cell_voxel, int ncells, int cap,
           const float* __restrict__ ymm3, const float* __restrict__ bn3, float* __restrict__ grid,
           float* __restrict__ vout, float* __restrict__ delta) {
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const bool has_empty = ncells - V > 0;                 // otherwise row V of ymm3 was never written
    const int q = threadIdx.x & 15;
    const float4 sc = reinterpret_cast<const float4*>(bn3)[q];
    const float4 sh = reinterpret_cast<const float4*>(bn3 + 64)[q];
    auto value = [&](int v) {
        const float4 mx = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128)[q];
        const float4 mn = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128 + 64)[q];
        return make_float4(pool_from(mx.x, mn.x, sc.x, sh.x), pool_from(mx.y, mn.y, sc.y, sh.y),
                           pool_from(mx.z, mn.z, sc.z, sh.z), pool_from(mx.w, mn.w, sc.w, sh.w));
    };
    const float4 c = has_empty ? value(V) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long long total = (long long)ncells * 16;
    const long long nv = vout ? (long long)(V + 1) * 16 : 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cell = (int)(i >> 4);
        const int v = cell_voxel[cell];
        reinterpret_cast<float4*>(grid)[i] = v < 0 ? c : value(v);
        if (i < nv) {
            const int u = (int)(i >> 4);
            const float4 o = u < V ? value(u) : c;
            reinterpret_cast<float4*>(vout)[i] = o;
            reinterpret_cast<float4*>(delta)[i] = make_float4(o.x - c.x, o.y - c.y, o.z - c.z, o.w - c.w);
        }
    }
    if (vout && !has_empty && blockIdx.x == 0 && threadIdx.x < 16) {
        // every cell occupied: row V (the "empty cell" constant) lies beyond the loop above and is defined as 0
        reinterpret_cast<float4*>(vout)[(size_t)V * 16 + threadIdx.x] = c;
        reinterpret_cast<float4*>(delta)[(size_t)V * 16 + threadIdx.x] = c;
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

// three partial tables (one per layer: stage L+1 reads table L while it writes table L+1)
extern "C" size_t lisec_vfe_workspace_bytes(void) {
    return 3 * align_up(sizeof(double) * (size_t)kFwdBlocks * 2 * 64, 256);
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
    Carver carve(workspace);
    double* parts1 = carve.take<double>((size_t)kFwdBlocks * 2 * 64);
    double* parts2 = carve.take<double>((size_t)kFwdBlocks * 2 * 64);
    double* parts3 = carve.take<double>((size_t)kFwdBlocks * 2 * 64);
    const double N = (double)ncells * (double)T;        // dense rows Keras reduces over (B = 1)
    StageBn bn;
    float* saved_bn[3] = {sv.bn1, sv.bn2, sv.bn3};
    for (int i = 0; i < 3; ++i) {
        bn.gamma[i] = p->gamma[i]; bn.beta[i] = p->beta[i];
        bn.mmean[i] = p->moving_mean[i]; bn.mvar[i] = p->moving_var[i];
        bn.saved[i] = saved_bn[i];
    }
    dim3 g(kFwdBlocks), b(kFwdThreads);
    if (training) {
        hipLaunchKernelGGL(k_vfe_stage<1>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2], bn, N,
                           sv.ymm1, sv.ymm2, sv.ymm3, (const double*)nullptr, parts1);
        hipLaunchKernelGGL(k_vfe_stage<2>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2], bn, N,
                           sv.ymm1, sv.ymm2, sv.ymm3, (const double*)parts1, parts2);
        hipLaunchKernelGGL(k_vfe_stage<3>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2], bn, N,
                           sv.ymm1, sv.ymm2, sv.ymm3, (const double*)parts2, parts3);
        hipLaunchKernelGGL(k_vfe_final3, dim3(64), dim3(kFwdBlocks), 0, st, (const double*)parts3, N, p->gamma[2],
                           p->beta[2], p->moving_mean[2], p->moving_var[2], sv.bn3);
        LISEC_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_vfe_stage<0>, g, b, 0, st, in, p->kernel[0], p->kernel[1], p->kernel[2], bn, N,
                           sv.ymm1, sv.ymm2, sv.ymm3, (const double*)nullptr, (double*)nullptr);
        LISEC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_vfe_grid, dim3(4096), dim3(256), 0, st, info, cell_voxel, ncells, cap_voxels,
                       sv.ymm3, sv.bn3, grid, sv.vout, sv.delta);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_vfe_grid_from_saved(const int32_t* info, const int32_t* cell_voxel, int ncells,
                                         int cap_voxels, const float* saved, float* grid, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(info && cell_voxel && saved && grid && ncells > 0 && cap_voxels >= 0, "bad arguments");
    VfeSaved sv(const_cast<float*>(saved), cap_voxels);
    hipLaunchKernelGGL(k_vfe_grid, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream_), info, cell_voxel,
                       ncells, cap_voxels, sv.ymm3, sv.bn3, grid, (float*)nullptr, (float*)nullptr);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
