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
// Training needs the batch statistics of a layer before the next layer can run => a launch boundary
// per BatchNormalization -- except the first one, whose statistics are a closed form of the row moments
// the voxeliser leaves in row_stats (Dense(6,16) has no bias and only the input in front of it).  The
// statistics of layers 2 and 3 are summed across workgroups in two-limb fixed-point integer accumulators
// (order-independent, hence deterministic) and finalised in the PROLOGUE of the consuming kernel by every
// workgroup (workgroup 0 also stores bnstate and updates the moving statistics): 3 launches -- layers
// 1+2, layer 3, grid writer.  Inference (moving statistics) runs the three passes in one kernel: 2 launches.  Voxel metadata and input rows are prefetched one and two
// voxels ahead (the stage kernels are latency-, not throughput-bound on a 20 k-point sweep).  The
// last kernel streams the dense (D,H,W,64) grid: 164 MB for the Lyft grid, the HBM-bound part of this
// file (empty cells hold the non-zero constant relu(BN3(.)), so the write is compulsory), and writes
// the compact per-voxel outputs (vout / delta) on the side.
#include "vfe_common.h"


namespace lisec {
namespace {

// Diagnostic only (tools/vfe_stamps.py): 100 MHz s_memrealtime stamps of wave 0 of every workgroup at the phase
// boundaries of the stage kernels, written to a buffer of their own; NULL (the default) = no stamp executes.
__device__ unsigned long long* g_vfe_stamps = nullptr;
#define LISEC_STAMP(K_)                                                                                         \
    do {                                                                                                        \
        if (stamps && threadIdx.x == 0)                                                                         \
            stamps[((size_t)(STAGE == 3) * 4096 + blockIdx.x) * 8 + (K_)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)

constexpr int kAccReplicas = 4;      // replicas of the cross-workgroup statistic accumulators (atomic contention)

// Cross-workgroup statistic accumulators: long long[kAccReplicas][2*C][2] two-limb fixed-point sums (common.h):
// integer atomics commute, so the totals -- and everything derived from them -- are deterministic.
template <int C>
__device__ __forceinline__ void acc_add(long long* acc, int q, int c, double v) {
    fx_atomic_add(acc + ((size_t)(blockIdx.x % kAccReplicas) * 2 * C + q * C + c) * 2, v);
}
template <int C>
__device__ __forceinline__ double acc_read(const long long* acc, int q, int c) {
    long long hi = 0, lo = 0;
#pragma unroll
    for (int r = 0; r < kAccReplicas; ++r) {
        const long long* p = acc + ((size_t)r * 2 * C + q * C + c) * 2;
        hi += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return fx_join(hi, lo);
}

// BatchNormalization state of one channel from (sum y, sum y^2) over N dense rows: scale / shift into LDS; workgroup 0
// also stores the bnstate the backward reads and updates the moving statistics (biased batch variance: the rank-6 VFE
// path of Keras is not the fused one).
template <int C>
__device__ __forceinline__ void bn_from_sums(int c, double s1, double s2, double N, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float* __restrict__ mmean,
                                             float* __restrict__ mvar, float* __restrict__ bnsaved, float* sbn) {
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

// Moments of the feature rows for callers that did not get them from the voxeliser (row_stats == NULL):
// zero, then accumulate (same layout and arithmetic as voxelize.hip's k_features).
__global__ void k_vfe_stats_zero(long long* __restrict__ stats) {
    for (int i = threadIdx.x; i < LISEC_ROW_STATS_WORDS; i += blockDim.x) stats[i] = 0;
}
__global__ void __launch_bounds__(256)
k_vfe_stats(VfeIn in, long long* __restrict__ stats) {
    __shared__ double smom[4][27];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int R = in.info[LISEC_VI_NROWS];
    double mom[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) mom[k] = 0.0;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
        double f[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) f[k] = (double)in.rows[(size_t)r * 6 + k];
        int q = 6;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            mom[j] += f[j];
#pragma unroll
            for (int k = j; k < 6; ++k) mom[q++] += f[j] * f[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const double t = wave_sum(mom[k]);
        if (lane == 0) smom[w][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        const double t = ((smom[0][threadIdx.x] + smom[1][threadIdx.x]) + smom[2][threadIdx.x]) + smom[3][threadIdx.x];
        if (t != 0.0) fx_atomic_add(stats + ((size_t)(blockIdx.x % LISEC_ROW_STATS_REPLICAS) * 27 + threadIdx.x) * 2, t);
    }
}

// STAGE 2: training, layers 1 + 2.  BatchNormalization 1 needs no pass over the rows: Dense(6,16) has no bias and
//          nothing but the input in front of it, so (sum y, sum y^2) of channel c are w_c . m and w_c^T S w_c with the
//          row moments m, S the voxeliser left in row_stats (pad rows and empty voxels add 0).  The batch statistics
//          of layer 2 go to the accumulators acc_out.
// STAGE 3: training, layer 3: statistics of layer 2 read from acc_in, those of layer 3 added to acc_out.
// STAGE 0: inference, all three layers, statistics from the moving averages.
template <int STAGE, int kFwdWaves, bool LDSW>
__global__ void __launch_bounds__(kFwdWaves * 64)
k_vfe_stage(VfeIn in, const float* __restrict__ W1, const float* __restrict__ W2,
            const float* __restrict__ W3, StageBn bn, double N,
            float* __restrict__ ymm1, float* __restrict__ ymm2, float* __restrict__ ymm3,
            const long long* __restrict__ row_stats, const long long* __restrict__ acc_in,
            long long* __restrict__ acc_out, long long* __restrict__ acc_zero,
            unsigned char* __restrict__ arg1, unsigned char* __restrict__ arg2, unsigned char* __restrict__ arg3,
            float* __restrict__ y2rows) {
    __shared__ float sbn1[32], sbn2[64];
    __shared__ double red[2 * kFwdWaves * 64];
    constexpr int kFwdThreads = kFwdWaves * 64;
    __shared__ float sW2p[LDSW ? 16 * 32 : 1], sW3p[(LDSW && STAGE != 2) ? 32 * 64 : 1];   // halves for the pooled inputs
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c1 = lane & 15, c2 = lane & 31;
    unsigned long long* stamps = g_vfe_stamps;
    LISEC_STAMP(0);
    // everything the first voxels need is requested before the statistics are finalised -- and before the voxel count V
    // is known: npts / row_start hold `cap` entries, so the speculative reads are in bounds and are masked by V below
    int V = in.info[LISEC_VI_NVOX];
    VfeWeights W;
    W.load(W1, W2, W3, STAGE, /*pooled=*/!LDSW);
    const int nwaves = gridDim.x * kFwdWaves;
    auto load_meta_spec = [&](int v, int& s, int& rs) {
        s = 0; rs = 0;
        if (v < in.cap) { s = in.npts[v]; rs = in.row_start[v]; }
    };
    int v = blockIdx.x * kFwdWaves + w;
    int s_cur, rs_cur, s_nxt, rs_nxt;
    load_meta_spec(v, s_cur, rs_cur);
    load_meta_spec(v + nwaves, s_nxt, rs_nxt);
    if (LDSW) {
        for (int i = threadIdx.x; i < 16 * 32; i += kFwdThreads) sW2p[i] = W2[i];
        if (STAGE != 2)
            for (int i = threadIdx.x; i < 32 * 64; i += kFwdThreads) sW3p[i] = W3[i];
    }
    if (V > in.cap) V = in.cap;
    const int nE = in.ncells - V;
    const int nvox = V + (nE > 0 ? 1 : 0);
    if (v >= V) { s_cur = 0; rs_cur = 0; }
    if (v + nwaves >= V) { s_nxt = 0; rs_nxt = 0; }
    auto load_meta = [&](int v, int& s, int& rs) {
        s = 0; rs = 0;
        if (v < V) { s = in.npts[v]; rs = in.row_start[v]; }
    };
    LISEC_STAMP(1);
    if (STAGE == 0) {
        block_fold<16>(bn.gamma[0], bn.beta[0], bn.mmean[0], bn.mvar[0], bn.saved[0], sbn1);
        block_fold<32>(bn.gamma[1], bn.beta[1], bn.mmean[1], bn.mvar[1], bn.saved[1], sbn2);
    } else if (STAGE == 2) {
        // row moments -> LDS (27 doubles), then the closed form per channel
        if (threadIdx.x < 27) {
            long long hi = 0, lo = 0;
#pragma unroll
            for (int r = 0; r < LISEC_ROW_STATS_REPLICAS; ++r) {
                const long long* p = row_stats + ((size_t)r * 27 + threadIdx.x) * 2;
                hi += p[0]; lo += p[1];
            }
            red[threadIdx.x] = fx_join(hi, lo);
        }
        if (blockIdx.x == 0 && acc_zero)                 // the accumulators of the NEXT stage start at zero
            for (int i = threadIdx.x; i < kAccReplicas * 2 * 64 * 2; i += kFwdThreads) acc_zero[i] = 0;
        __syncthreads();
        if (threadIdx.x < 16) {
            const int c = threadIdx.x;
            double wv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) wv[k] = (double)W1[k * 16 + c];
            double s1 = 0.0, s2 = 0.0;
            int q = 6;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                s1 += wv[j] * red[j];
#pragma unroll
                for (int k = j; k < 6; ++k) { s2 += (j == k ? 1.0 : 2.0) * wv[j] * wv[k] * red[q]; ++q; }
            }
            bn_from_sums<16>(c, s1, s2, N, bn.gamma[0], bn.beta[0], bn.mmean[0], bn.mvar[0], bn.saved[0], sbn1);
        }
    } else if (STAGE == 3) {
        if (threadIdx.x < 32) sbn1[threadIdx.x] = bn.saved[0][threadIdx.x];      // written by stage 2's workgroup 0
        if (threadIdx.x < 32) {
            const int c = threadIdx.x;
            bn_from_sums<32>(c, acc_read<32>(acc_in, 0, c), acc_read<32>(acc_in, 1, c), N, bn.gamma[1], bn.beta[1],
                             bn.mmean[1], bn.mvar[1], bn.saved[1], sbn2);
        }
    }
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
        if (STAGE == 3 && v < nvox) {
            mx1 = ymm1[(size_t)v * 32 + c1]; mn1 = ymm1[(size_t)v * 32 + 16 + c1];
            mx2 = ymm2[(size_t)v * 64 + c2]; mn2 = ymm2[(size_t)v * 64 + 32 + c2];
        }
    };
    float xr[6], pmx1, pmn1, pmx2, pmn2;
    load_rows(s_cur, rs_cur, xr);
    load_ymm(v, pmx1, pmn1, pmx2, pmn2);
    LISEC_STAMP(2);
    __syncthreads();
    LISEC_STAMP(3);
    float sc1 = sbn1[c1], sh1 = sbn1[16 + c1], sc2 = 0, sh2 = 0;
    if (STAGE == 0 || STAGE == 3) { sc2 = sbn2[c2]; sh2 = sbn2[32 + c2]; }
    // the pad row after layer 1 is the same everywhere: relu(BN1(0)) = relu(shift1)
    const float a1pad = fmaxf(sh1, 0.0f);
    float A2pad = 0.0f;                                     // a1pad @ W2[16:, :]
#pragma unroll
    for (int k = 0; k < 16; ++k) A2pad = fmaf(rl(a1pad, k), W.w2a[k], A2pad);
    double s1 = 0.0, s2 = 0.0;
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
        // ---- pass 1: y1 = x @ W1 (per-voxel max / min only) ------------------------------------
        if (STAGE == 0 || STAGE == 2) {
            mx1 = has_pad ? 0.0f : -INFINITY;               // pad row: 0 @ W1 == 0
            mn1 = has_pad ? 0.0f : INFINITY;
            int ax = 0, an = 0;                              // slot of the first row holding the max / min (0 = pad row)
            for (int t = 0; t < s; ++t) {
                float y = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) y = fmaf(rl(xr[k], t), W.w1[k], y);
                if (y > mx1) ax = t + 1;
                if (y < mn1) an = t + 1;
                mx1 = fmaxf(mx1, y); mn1 = fminf(mn1, y);
            }
            if (STAGE == 2 && lane < 16) {
                ymm1[(size_t)v * 32 + lane] = mx1; ymm1[(size_t)v * 32 + 16 + lane] = mn1;
                if (arg1) { arg1[(size_t)v * 32 + lane] = (unsigned char)ax; arg1[(size_t)v * 32 + 16 + lane] = (unsigned char)an; }
            }
        }
        // ---- pass 2: y2 = [pool1, a1] @ W2 -----------------------------------------------------
        const float pool1 = pool_from(mx1, mn1, sc1, sh1);
        float P2 = 0.0f;
        // LDSW: read from LDS once per voxel; the opaque zero keeps the compiler from hoisting the 48 reads back into
        // registers while still letting it batch them inside the iteration
        int opq = 0;
        if (LDSW) asm volatile("" : "+v"(opq));
#pragma unroll
        for (int k = 0; k < 16; ++k) P2 = fmaf(rl(pool1, k), LDSW ? sW2p[k * 32 + c2 + opq] : W.w2p[k], P2);
        if (STAGE == 0 || STAGE == 2) {
            const float y2pad = P2 + A2pad;
            mx2 = has_pad ? y2pad : -INFINITY;
            mn2 = has_pad ? y2pad : INFINITY;
            int ax = 0, an = 0;
            const size_t slot0 = (size_t)(virt ? in.info[LISEC_VI_NROWS] : rs_cur) + v;     // row_start[v] + v
            if (STAGE == 2 && has_pad) { s1 += wpad * (double)y2pad; s2 += wpad * (double)y2pad * (double)y2pad; }
            if (STAGE == 2 && y2rows && lane < 32) y2rows[slot0 * 32 + lane] = y2pad;
            for (int t = 0; t < s; ++t) {
                float y1 = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) y1 = fmaf(rl(xr[k], t), W.w1[k], y1);
                const float a1 = bnrelu(y1, sc1, sh1);
                float y = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; ++k) y = fmaf(rl(a1, k), W.w2a[k], y);
                y += P2;
                if (y > mx2) ax = t + 1;
                if (y < mn2) an = t + 1;
                mx2 = fmaxf(mx2, y); mn2 = fminf(mn2, y);
                if (STAGE == 2) { s1 += (double)y; s2 += (double)y * (double)y; }
                if (STAGE == 2 && y2rows && lane < 32) y2rows[(slot0 + t + 1) * 32 + lane] = y;
            }
            if (STAGE == 2 && lane < 32) {
                ymm2[(size_t)v * 64 + lane] = mx2; ymm2[(size_t)v * 64 + 32 + lane] = mn2;
                if (arg2) { arg2[(size_t)v * 64 + lane] = (unsigned char)ax; arg2[(size_t)v * 64 + 32 + lane] = (unsigned char)an; }
            }
        }
        if (STAGE != 2) {
            // ---- pass 3: y3 = [pool2, a2] @ W3 -------------------------------------------------
            const float pool2 = pool_from(mx2, mn2, sc2, sh2);
            float P3 = 0.0f;
#pragma unroll
            for (int k = 0; k < 32; ++k) P3 = fmaf(rl(pool2, k), LDSW ? sW3p[k * 64 + lane + opq] : W.w3p[k], P3);
            float mx3 = -INFINITY, mn3 = INFINITY;
            int ax3 = 0, an3 = 0;
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
                if (y > mx3) ax3 = t + 1;
                if (y < mn3) an3 = t + 1;
                mx3 = fmaxf(mx3, y); mn3 = fminf(mn3, y);
                if (STAGE == 3) { s1 += (double)y; s2 += (double)y * (double)y; }
            }
            ymm3[(size_t)v * 128 + lane] = mx3;
            ymm3[(size_t)v * 128 + 64 + lane] = mn3;
            if (STAGE == 3 && arg3) { arg3[(size_t)v * 128 + lane] = (unsigned char)ax3; arg3[(size_t)v * 128 + 64 + lane] = (unsigned char)an3; }
        }
        s_cur = s_nxt; rs_cur = rs_nxt; s_nxt = s_n2; rs_nxt = rs_n2;
#pragma unroll
        for (int k = 0; k < 6; ++k) xr[k] = xn[k];
        pmx1 = nmx1; pmn1 = nmn1; pmx2 = nmx2; pmn2 = nmn2;
        if (v == (int)(blockIdx.x * kFwdWaves + w)) LISEC_STAMP(4);        // end of this wave's first voxel
    }
    LISEC_STAMP(5);
    if (STAGE != 0) {
        constexpr int C = STAGE == 2 ? 32 : 64;
        __syncthreads();                                     // `red` was used by the prologue
        red[(0 * kFwdWaves + w) * 64 + lane] = s1;
        red[(1 * kFwdWaves + w) * 64 + lane] = s2;
        __syncthreads();
        if (threadIdx.x < 2 * C) {
            const int q = threadIdx.x / C, c = threadIdx.x % C;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < kFwdWaves; ++k) a += red[(q * kFwdWaves + k) * 64 + c];
            if (a != 0.0) acc_add<C>(acc_out, q, c, a);
        }
        LISEC_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LISEC_STAMP(7);
    } else if (blockIdx.x == 0) {
        // inference: the grid writer reads bn3 from `saved`
        __shared__ float sbn3[128];
        block_fold<64>(bn.gamma[2], bn.beta[2], bn.mmean[2], bn.mvar[2], bn.saved[2], sbn3);
    }
}

// Dense (ncells, 64) grid: occupied cells from their voxel's ymm3, empty cells from the virtual voxel.
// acc3 != NULL (training): the statistics of layer 3 are finalised here, by every workgroup, from the accumulators
// stage 3 filled (workgroup 0 stores bnstate / moving statistics and re-zeroes stage 2's accumulators, acc2_zero);
// acc3 == NULL: bn3 is read from `saved` (inference, or a rewrite of the grid).
// On the side (vout != NULL) the compact per-voxel outputs: vout[v] = the grid value of voxel v (v == V: the
// constant every empty cell holds, 0 when the grid has no empty cell), delta[v] = vout[v] - vout[V] (what the
// first Conv3D's sparse backward contracts against).
__global__ void __launch_bounds__(256)
k_vfe_grid(const int* __restrict__ info, const int* __restrict__ cell_voxel, int ncells, int cap,
           const float* __restrict__ ymm3, float* __restrict__ bn3, float* __restrict__ grid,
           float* __restrict__ vout, float* __restrict__ delta, const long long* __restrict__ acc3, double N,
           const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mmean,
           float* __restrict__ mvar, long long* __restrict__ acc2_zero) {
    __shared__ float sbn3[128];
    __shared__ double ssum[128];
    if (acc3) {
        if (threadIdx.x < 128) ssum[threadIdx.x] = acc_read<64>(acc3, threadIdx.x >> 6, threadIdx.x & 63);
        __syncthreads();
        if (threadIdx.x < 64)
            bn_from_sums<64>(threadIdx.x, ssum[threadIdx.x], ssum[64 + threadIdx.x], N, gamma, beta, mmean, mvar, bn3, sbn3);
        if (blockIdx.x == 0 && acc2_zero)
            for (int i = threadIdx.x; i < kAccReplicas * 2 * 32 * 2; i += 256) acc2_zero[i] = 0;
    } else if (threadIdx.x < 128) {
        sbn3[threadIdx.x] = bn3[threadIdx.x];
    }
    __syncthreads();
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const bool has_empty = ncells - V > 0;                 // otherwise row V of ymm3 was never written
    const int q = threadIdx.x & 15;
    const float4 sc = reinterpret_cast<const float4*>(sbn3)[q];
    const float4 sh = reinterpret_cast<const float4*>(sbn3 + 64)[q];
    auto value = [&](int v) {
        const float4 mx = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128)[q];
        const float4 mn = reinterpret_cast<const float4*>(ymm3 + (size_t)v * 128 + 64)[q];
        return make_float4(pool_from(mx.x, mn.x, sc.x, sh.x), pool_from(mx.y, mn.y, sc.y, sh.y),
                           pool_from(mx.z, mn.z, sc.z, sh.z), pool_from(mx.w, mn.w, sc.w, sh.w));
    };
    const float4 c = has_empty ? value(V) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long long nv = vout ? (long long)(V + 1) * 16 : 0;
    // grid == nullptr: the compact per-voxel outputs only (what the field form of the first Conv3D reads)
    const long long total = grid ? (long long)ncells * 16 : (nv < (long long)ncells * 16 ? nv : (long long)ncells * 16);
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (grid) {
            const int v = cell_voxel[(int)(i >> 4)];
            reinterpret_cast<float4*>(grid)[i] = v < 0 ? c : value(v);
        }
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

extern "C" size_t lisec_vfe_saved_floats_rows(int cap_voxels, int n_points) {
    if (cap_voxels < 0 || n_points < 0) return 0;
    return VfeSaved(nullptr, cap_voxels, n_points).floats;
}

// the accumulators of layer 3 (+ a row_stats block for callers that pass none)
extern "C" size_t lisec_vfe_workspace_bytes(void) {
    return align_up(sizeof(long long) * (size_t)kAccReplicas * 2 * 64 * 2, 256) +
           align_up(sizeof(long long) * (size_t)LISEC_ROW_STATS_WORDS, 256);
}

extern "C" int lisec_vfe_forward(const lisec_vfe_params* p, const int32_t* info,
                                 const int32_t* cell_voxel, const int32_t* npts,
                                 const int32_t* row_start, const float* rows, int64_t* row_stats_, int n_points,
                                 int ncells, int T, int cap_voxels, int training, float* saved, void* workspace,
                                 size_t workspace_bytes, float* grid, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(p && info && cell_voxel && npts && row_start && rows && saved && workspace, "NULL pointer");
    LISEC_CHECK_ARG(ncells > 0 && T >= 1 && T <= 64 && cap_voxels >= 0 && n_points >= 0, "bad sizes");
    for (int i = 0; i < 3; ++i)
        LISEC_CHECK_ARG(p->kernel[i] && p->gamma[i] && p->beta[i] && p->moving_mean[i] && p->moving_var[i],
                        "NULL VFE parameter pointer");
    if (workspace_bytes < lisec_vfe_workspace_bytes()) {
        set_error("vfe workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    VfeSaved sv(saved, cap_voxels, training ? n_points : 0);       // n_points > 0: `saved` carries the per-row extras
    VfeIn in{info, npts, row_start, rows, ncells, T, cap_voxels};
    Carver carve(workspace);
    long long* acc3 = carve.take<long long>((size_t)kAccReplicas * 2 * 64 * 2);
    long long* own_stats = carve.take<long long>(LISEC_ROW_STATS_WORDS);
    const double N = (double)ncells * (double)T;        // dense rows Keras reduces over (B = 1)
    StageBn bn;
    float* saved_bn[3] = {sv.bn1, sv.bn2, sv.bn3};
    for (int i = 0; i < 3; ++i) {
        bn.gamma[i] = p->gamma[i]; bn.beta[i] = p->beta[i];
        bn.mmean[i] = p->moving_mean[i]; bn.mvar[i] = p->moving_var[i];
        bn.saved[i] = saved_bn[i];
    }
    // launch shape of the stage kernels.  Measured on a 20 k-point sweep (tools/bench_vfe.py, tools/vfe_stamps.py): 256
    // or 512 workgroups of 8 waves, pooled-input weights in registers or LDS -- all within 57.7-58.8 us for the whole
    // call; fewer (128) or smaller (4-, 2-wave) workgroups are slower.  Default: one 8-wave workgroup per CU, LDS weights;
    // two per CU for big sweeps (84 000 voxels: 253 -> 215 us for the call; nothing to choose between them at 9 400).
    const int shape_set = tuning().vfe_shape;
    const int shape = shape_set >= 0 ? shape_set : (cap_voxels > 32768 ? 5 : 1);
#define LISEC_STAGE(ST_, ...)                                                                                   \
    do {                                                                                                        \
        if (shape == 0) LISEC_LAUNCH((k_vfe_stage<ST_, 8, false>), dim3(256), dim3(512), 0, st, __VA_ARGS__);      \
        else if (shape == 5) LISEC_LAUNCH((k_vfe_stage<ST_, 8, true>), dim3(512), dim3(512), 0, st, __VA_ARGS__);  \
        else LISEC_LAUNCH((k_vfe_stage<ST_, 8, true>), dim3(256), dim3(512), 0, st, __VA_ARGS__);                  \
    } while (0)
    if (training) {
        long long* stats = reinterpret_cast<long long*>(row_stats_);
        if (!stats) {                                   // no moments from the voxeliser: sum them here
            stats = own_stats;
            LISEC_LAUNCH(k_vfe_stats_zero, dim3(1), dim3(1024), 0, st, stats);
            LISEC_LAUNCH(k_vfe_stats, dim3(256), dim3(256), 0, st, in, stats);
        }
        long long* acc2 = stats + LISEC_ROW_STATS_MOMENT_WORDS;       // zero on entry, re-zeroed by the grid writer
        LISEC_STAGE(2, in, p->kernel[0], p->kernel[1], p->kernel[2], bn, N, sv.ymm1, sv.ymm2, sv.ymm3,
                    (const long long*)stats, (const long long*)nullptr, acc2, acc3, sv.arg1, sv.arg2, sv.arg3, sv.y2rows);
        LISEC_STAGE(3, in, p->kernel[0], p->kernel[1], p->kernel[2], bn, N, sv.ymm1, sv.ymm2, sv.ymm3,
                    (const long long*)nullptr, (const long long*)acc2, acc3, (long long*)nullptr, sv.arg1, sv.arg2,
                    sv.arg3, sv.y2rows);
        LISEC_LAUNCH_CHECK();
        LISEC_LAUNCH(k_vfe_grid, dim3(2048), dim3(256), 0, st, info, cell_voxel, ncells, cap_voxels,
                           sv.ymm3, sv.bn3, grid, sv.vout, sv.delta, (const long long*)acc3, N, p->gamma[2],
                           p->beta[2], p->moving_mean[2], p->moving_var[2], acc2);
    } else {
        LISEC_STAGE(0, in, p->kernel[0], p->kernel[1], p->kernel[2], bn, N, sv.ymm1, sv.ymm2, sv.ymm3,
                    (const long long*)nullptr, (const long long*)nullptr, (long long*)nullptr, (long long*)nullptr,
                    (unsigned char*)nullptr, (unsigned char*)nullptr, (unsigned char*)nullptr, (float*)nullptr);
        LISEC_LAUNCH_CHECK();
        LISEC_LAUNCH(k_vfe_grid, dim3(2048), dim3(256), 0, st, info, cell_voxel, ncells, cap_voxels,
                           sv.ymm3, sv.bn3, grid, sv.vout, sv.delta, (const long long*)nullptr, N,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (long long*)nullptr);
    }
#undef LISEC_STAGE
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_vfe_grid_from_saved(const int32_t* info, const int32_t* cell_voxel, int ncells,
                                         int cap_voxels, const float* saved, float* grid, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(info && cell_voxel && saved && grid && ncells > 0 && cap_voxels >= 0, "bad arguments");
    VfeSaved sv(const_cast<float*>(saved), cap_voxels);
    LISEC_LAUNCH(k_vfe_grid, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream_), info, cell_voxel,
                       ncells, cap_voxels, sv.ymm3, sv.bn3, grid, (float*)nullptr, (float*)nullptr,
                       (const long long*)nullptr, 0.0, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (long long*)nullptr);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

// Diagnostic (not in lisec_hip.h): points the stage kernels' stamp buffer at `buf` (device, 2*4096*8 uint64) or NULL.
extern "C" int lisec_debug_vfe_stamps(unsigned long long* buf) {
    LISEC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_vfe_stamps), &buf, sizeof(buf)));
    return LISEC_OK;
}
