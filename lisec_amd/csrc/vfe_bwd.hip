// Sparse-exact VFE stack for gfx950 -- backward (training-mode BatchNormalization).
//
// Gradients Keras' fit() derives for the VFE variables (Dense kernels + BN gamma/beta of
// addVFELayer(6,32), addVFELayer(32,64), addFCN(64,64); model_training.py:155-186, :231-235, :299)
// from the gradient of the dense (D,H,W,64) grid.  Same row-class formulation as the forward
// (vfe.hip; derivation and proof against dense autograd: oracle/vfe_sparse_ref.py): every class
// representative carries the SUM of the gradients of its identical dense copies, which is exact
// because everything below a max is linear in the gradient; reduce_max routes the gradient to the
// first row attaining the maximum (ties only occur between identical copies or at relu-clamped
// zeros, where the gradient vanishes either way).
//
//   k_gather        dgrid -> dout[v] (occupied cells) + sum over empty cells (the virtual voxel)
//   k_l3_stats      dbeta3/dgamma3 from the winner rows (they are the rows holding ymax/ymin)
//   k_l3            per voxel rows: GY3 -> dW3, GH = GY3 @ W3^T -> pooled half to the layer-2 winner,
//                   pointwise half per row -> GZ2 rows (HBM, 128 B/row) + dbeta2/dgamma2 partials
//   k_l2            same one layer down -> GZ1 rows + dbeta1/dgamma1 partials, dW2
//   k_l1            dW1
// One wave per voxel, lane = channel, activations recomputed from the 24-byte input rows; LDS only
// holds the current voxel's per-row scratch.  Reductions are per-block partials in fp64 summed in
// index order (deterministic).
#include "vfe_common.h"

namespace lisec {
namespace {

constexpr int kBwdBlocks = 512;
constexpr int kMaxRows = 64;            // rows of one voxel (<= T <= 64)

struct Vox {
    int s, rs, nrows;
    bool has_pad;
    float wpad;
    float xr[6];
};

__device__ __forceinline__ Vox load_vox(const VfeIn& in, int v, int V, int nE) {
    Vox x;
    const bool virt = v == V;
    x.s = virt ? 0 : in.npts[v];
    x.rs = virt ? 0 : in.row_start[v];
    x.has_pad = virt || x.s < in.T;
    x.wpad = virt ? (float)in.T * (float)nE : (float)(in.T - x.s);
    x.nrows = x.s + (x.has_pad ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 6; ++k) x.xr[k] = 0.f;
    const int lane = lane_id();
    if (lane < x.s) {
#pragma unroll
        for (int k = 0; k < 6; ++k) x.xr[k] = in.rows[(size_t)(x.rs + lane) * 6 + k];
    }
    return x;
}

// index of class row j of voxel v in the per-row scratch arrays (slot 0 of every voxel = its pad row)
__device__ __forceinline__ size_t crow(const VfeIn& in, int v, int V, int j, bool has_pad) {
    const int rs = in.row_start[v];             // row_start[V] = total real rows (voxeliser header)
    return (size_t)rs + v + (has_pad ? j : j + 1);
}

__global__ void __launch_bounds__(256)
k_gather(const int* __restrict__ info, const int* __restrict__ cell_voxel, int ncells, int cap,
         const float* __restrict__ dgrid, float* __restrict__ dout, double* __restrict__ parts) {
    __shared__ float red[256][4];
    const int q = threadIdx.x & 15;
    float4 es = make_float4(0, 0, 0, 0);
    const long long total = (long long)ncells * 16;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cell = (int)(i >> 4);
        const int v = cell_voxel[cell];
        const float4 g = reinterpret_cast<const float4*>(dgrid)[i];
        if (v >= 0) {
            reinterpret_cast<float4*>(dout + (size_t)v * 64)[q] = g;
        } else {
            es.x += g.x; es.y += g.y; es.z += g.z; es.w += g.w;
        }
    }
    red[threadIdx.x][0] = es.x; red[threadIdx.x][1] = es.y; red[threadIdx.x][2] = es.z; red[threadIdx.x][3] = es.w;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        double a = 0.0;
        for (int k = 0; k < 16; ++k) a += (double)red[k * 16 + c / 4][c % 4];
        parts[(size_t)blockIdx.x * 64 + c] = a;
    }
}

__global__ void __launch_bounds__(1024)
k_virtual_dout(const int* __restrict__ info, int cap, const double* __restrict__ parts, int nparts,
               float* __restrict__ dout) {
    __shared__ double red[16][64];
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const int c = threadIdx.x & 63, ry = threadIdx.x >> 6;
    double a = 0.0;
    for (int b = ry; b < nparts; b += 16) a += parts[(size_t)b * 64 + c];
    red[ry][c] = a;
    __syncthreads();
    for (int o = 8; o > 0; o >>= 1) {
        if (ry < o) red[ry][c] += red[ry + o][c];
        __syncthreads();
    }
    if (ry == 0) dout[(size_t)V * 64 + c] = (float)red[0][c];
}

// column sums of the compact gradient rows [0, V) as per-block fp64 parts (rows b, b+G, ... per block)
__global__ void __launch_bounds__(256)
k_rows_colsum(const int* __restrict__ info, int cap, const float* __restrict__ dout, double* __restrict__ parts) {
    __shared__ double red[4][64];
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const int c = threadIdx.x & 63, ry = threadIdx.x >> 6;
    double a = 0.0;
    for (int v = blockIdx.x * 4 + ry; v < V; v += gridDim.x * 4) a += (double)dout[(size_t)v * 64 + c];
    red[ry][c] = a;
    __syncthreads();
    if (ry == 0) parts[(size_t)blockIdx.x * 64 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// virtual voxel: (sum of the grid gradient over ALL cells) - (sum over the occupied cells)
__global__ void __launch_bounds__(1024)
k_virtual_from_total(const int* __restrict__ info, int cap, const double* __restrict__ parts, int nparts,
                     const float* __restrict__ g_all, float* __restrict__ dout) {
    __shared__ double red[16][64];
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const int c = threadIdx.x & 63, ry = threadIdx.x >> 6;
    double a = 0.0;
    for (int b = ry; b < nparts; b += 16) a += parts[(size_t)b * 64 + c];
    red[ry][c] = a;
    __syncthreads();
    for (int o = 8; o > 0; o >>= 1) {
        if (ry < o) red[ry][c] += red[ry + o][c];
        __syncthreads();
    }
    if (ry == 0) dout[(size_t)V * 64 + c] = (float)((double)g_all[c] - red[0][c]);
}

__device__ __forceinline__ void block_stats_out(double s1, double s2, int C, double* parts, int w, int lane) {
    __shared__ double red[2][4][64];
    red[0][w][lane] = s1; red[1][w][lane] = s2;
    __syncthreads();
    if ((int)threadIdx.x < 2 * C) {
        const int q = threadIdx.x / C, c = threadIdx.x % C;
        parts[((size_t)blockIdx.x * 2 + q) * C + c] = red[q][0][c] + red[q][1][c] + red[q][2][c] + red[q][3][c];
    }
}

// dbeta3/dgamma3: only the row attaining the per-voxel max of a3 carries gradient, and that row's
// pre-BN value is ymax3 (scale >= 0) or ymin3 (scale < 0)
__global__ void __launch_bounds__(256)
k_l3_stats(VfeIn in, const float* __restrict__ bn3, const float* __restrict__ ymm3,
           const float* __restrict__ dout, double* __restrict__ parts) {
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nvox = V + (in.ncells - V > 0 ? 1 : 0);
    const float sc = bn3[lane], sh = bn3[64 + lane], mu = bn3[128 + lane], is = bn3[192 + lane];
    double s1 = 0.0, s2 = 0.0;
    for (int v = blockIdx.x * 4 + w; v < nvox; v += gridDim.x * 4) {
        const float ys = sc >= 0.f ? ymm3[(size_t)v * 128 + lane] : ymm3[(size_t)v * 128 + 64 + lane];
        const float gz = fmaf(ys, sc, sh) > 0.f ? dout[(size_t)v * 64 + lane] : 0.f;
        s1 += (double)gz;
        s2 += (double)gz * (double)((ys - mu) * is);
    }
    block_stats_out(s1, s2, 64, parts, w, lane);
}

// sequential block reduction of per-wave accumulators acc[K] (lane = column) into parts[blk][K][C]
template <int K>
__device__ __forceinline__ void block_dw_out(const float (&acc)[K], int C, float* lds, double* parts, int w, int lane) {
    for (int round = 0; round < 4; ++round) {
        __syncthreads();
        if (w == round && lane < C) {
#pragma unroll
            for (int k = 0; k < K; ++k) lds[k * C + lane] = (round == 0 ? 0.f : lds[k * C + lane]) + acc[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * C; i += 256) parts[(size_t)blockIdx.x * K * C + i] = (double)lds[i];
}

__global__ void __launch_bounds__(256)
k_l3(VfeIn in, const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ W3,
     const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn3,
     const float* __restrict__ coef3, const float* __restrict__ ymm1, const float* __restrict__ ymm2,
     const float* __restrict__ ymm3, const float* __restrict__ dout, float* __restrict__ gz2buf,
     double* __restrict__ dw3_parts, double* __restrict__ st2_parts) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sGA = smem + w * (2 * kMaxRows * 32);          // [rows][32] pointwise gradient wrt a2
    float* sY2 = sGA + kMaxRows * 32;                     // [rows][32] y2
    float* sDW = smem + 4 * (2 * kMaxRows * 32);          // [64][64] block reduction scratch
    const int c1 = lane & 15, c2 = lane & 31;
    VfeWeights W;
    W.load(W1, W2, W3, 3);
    float w3row[64];                                       // W3[k = lane][c]
#pragma unroll
    for (int c = 0; c < 64; ++c) w3row[c] = W3[lane * 64 + c];
    const float sc1 = bn1[c1], sh1 = bn1[16 + c1];
    const float sc2 = bn2[c2], sh2 = bn2[32 + c2], mu2 = bn2[64 + c2], is2 = bn2[96 + c2];
    const float sc3 = bn3[lane], sh3 = bn3[64 + lane], mu3 = bn3[128 + lane], is3 = bn3[192 + lane];
    const float m1 = coef3[lane], m2 = coef3[64 + lane];
    const float a1pad = fmaxf(sh1, 0.f);
    float A2pad = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) A2pad = fmaf(rl(a1pad, k), W.w2a[k], A2pad);
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nE = in.ncells - V;
    const int nvox = V + (nE > 0 ? 1 : 0);
    float dwp[32], dwa[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { dwp[k] = 0.f; dwa[k] = 0.f; }
    double s1 = 0.0, s2 = 0.0;
    for (int v = blockIdx.x * 4 + w; v < nvox; v += gridDim.x * 4) {
        Vox x = load_vox(in, v, V, nE);
        const float pool1 = pool_from(ymm1[(size_t)v * 32 + c1], ymm1[(size_t)v * 32 + 16 + c1], sc1, sh1);
        float P2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) P2 = fmaf(rl(pool1, k), W.w2p[k], P2);
        const float ys2 = sc2 >= 0.f ? ymm2[(size_t)v * 64 + c2] : ymm2[(size_t)v * 64 + 32 + c2];
        const float pool2 = bnrelu(ys2, sc2, sh2);
        float P3 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) P3 = fmaf(rl(pool2, k), W.w3p[k], P3);
        const float ys3 = sc3 >= 0.f ? ymm3[(size_t)v * 128 + lane] : ymm3[(size_t)v * 128 + 64 + lane];
        const float dov = dout[(size_t)v * 64 + lane];
        bool found3 = false;
        float sgy = 0.f, gpool2 = 0.f;
        for (int j = 0; j < x.nrows; ++j) {
            const bool is_pad = x.has_pad && j == 0;
            const int t = x.has_pad ? j - 1 : j;
            const float wr = is_pad ? x.wpad : 1.f;
            float y1 = 0.f;
            if (!is_pad) {
#pragma unroll
                for (int k = 0; k < 6; ++k) y1 = fmaf(rl(x.xr[k], t), W.w1[k], y1);
            }
            const float a1 = bnrelu(y1, sc1, sh1);
            float y2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) y2 = fmaf(rl(a1, k), W.w2a[k], y2);
            y2 = is_pad ? P2 + A2pad : y2 + P2;
            const float a2 = bnrelu(y2, sc2, sh2);
            float y3 = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) y3 = fmaf(rl(a2, k), W.w3a[k], y3);
            y3 += P3;
            const bool hit = !found3 && y3 == ys3;
            found3 = found3 || hit;
            const float gz = (hit && fmaf(y3, sc3, sh3) > 0.f) ? dov : 0.f;
            const float yh = (y3 - mu3) * is3;
            const float gy = sc3 * (gz - wr * m1 - wr * yh * m2);
            sgy += gy;
#pragma unroll
            for (int k = 0; k < 32; ++k) dwa[k] = fmaf(rl(a2, k), gy, dwa[k]);
            float gh = 0.f;                                 // (GY3 @ W3^T)[k = lane]
#pragma unroll
            for (int c = 0; c < 64; ++c) gh = fmaf(rl(gy, c), w3row[c], gh);
            if (lane < 32) { gpool2 += gh; sY2[j * 32 + lane] = y2; }
            else sGA[j * 32 + lane - 32] = gh;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) dwp[k] = fmaf(rl(pool2, k), sgy, dwp[k]);
        __threadfence_block();
        bool found2 = false;
        for (int j = 0; j < x.nrows; ++j) {
            const bool is_pad = x.has_pad && j == 0;
            const float y2 = sY2[j * 32 + c2];
            const bool hit = !found2 && y2 == ys2;
            found2 = found2 || hit;
            float ga = sGA[j * 32 + c2] + (hit ? gpool2 : 0.f);
            if (!(fmaf(y2, sc2, sh2) > 0.f)) ga = 0.f;
            if (lane < 32) {
                gz2buf[crow(in, v, V, j, x.has_pad) * 32 + lane] = ga;
                s1 += (double)ga;
                s2 += (double)ga * (double)((y2 - mu2) * is2);
            }
            (void)is_pad;
        }
        __threadfence_block();
    }
    // dW3[k][c]: k < 32 pooled half, k >= 32 pointwise half
    float acc[64];
#pragma unroll
    for (int k = 0; k < 32; ++k) { acc[k] = dwp[k]; acc[32 + k] = dwa[k]; }
    block_dw_out<64>(acc, 64, sDW, dw3_parts, w, lane);
    block_stats_out(s1, s2, 32, st2_parts, w, lane);
}

__global__ void __launch_bounds__(256)
k_l2(VfeIn in, const float* __restrict__ W1, const float* __restrict__ W2, const float* __restrict__ bn1,
     const float* __restrict__ bn2, const float* __restrict__ coef2, const float* __restrict__ ymm1,
     const float* __restrict__ gz2buf, float* __restrict__ gz1buf, double* __restrict__ dw2_parts,
     double* __restrict__ st1_parts) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sGA = smem + w * (2 * kMaxRows * 16);          // [rows][16] pointwise gradient wrt a1
    float* sY1 = sGA + kMaxRows * 16;                     // [rows][16] y1
    float* sDW = smem + 4 * (2 * kMaxRows * 16);          // [32][32]
    const int c1 = lane & 15, c2 = lane & 31;
    VfeWeights W;
    W.load(W1, W2, nullptr, 2);
    float w2row[32];                                       // W2[k = lane&31][c]
#pragma unroll
    for (int c = 0; c < 32; ++c) w2row[c] = W2[c2 * 32 + c];
    const float sc1 = bn1[c1], sh1 = bn1[16 + c1], mu1 = bn1[32 + c1], is1 = bn1[48 + c1];
    const float sc2 = bn2[c2], mu2 = bn2[64 + c2], is2 = bn2[96 + c2];
    const float m1 = coef2[c2], m2 = coef2[32 + c2];
    const float a1pad = fmaxf(sh1, 0.f);
    float A2pad = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) A2pad = fmaf(rl(a1pad, k), W.w2a[k], A2pad);
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nE = in.ncells - V;
    const int nvox = V + (nE > 0 ? 1 : 0);
    float dwp[16], dwa[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { dwp[k] = 0.f; dwa[k] = 0.f; }
    double s1 = 0.0, s2 = 0.0;
    for (int v = blockIdx.x * 4 + w; v < nvox; v += gridDim.x * 4) {
        Vox x = load_vox(in, v, V, nE);
        const float ys1 = sc1 >= 0.f ? ymm1[(size_t)v * 32 + c1] : ymm1[(size_t)v * 32 + 16 + c1];
        const float pool1 = bnrelu(ys1, sc1, sh1);
        float P2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) P2 = fmaf(rl(pool1, k), W.w2p[k], P2);
        float sgy = 0.f, gpool1 = 0.f;
        for (int j = 0; j < x.nrows; ++j) {
            const bool is_pad = x.has_pad && j == 0;
            const int t = x.has_pad ? j - 1 : j;
            const float wr = is_pad ? x.wpad : 1.f;
            float y1 = 0.f;
            if (!is_pad) {
#pragma unroll
                for (int k = 0; k < 6; ++k) y1 = fmaf(rl(x.xr[k], t), W.w1[k], y1);
            }
            const float a1 = bnrelu(y1, sc1, sh1);
            float y2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) y2 = fmaf(rl(a1, k), W.w2a[k], y2);
            y2 = is_pad ? P2 + A2pad : y2 + P2;
            const float gz = gz2buf[crow(in, v, V, j, x.has_pad) * 32 + c2];
            const float yh = (y2 - mu2) * is2;
            const float gy = sc2 * (gz - wr * m1 - wr * yh * m2);
            sgy += gy;
#pragma unroll
            for (int k = 0; k < 16; ++k) dwa[k] = fmaf(rl(a1, k), gy, dwa[k]);
            float gh = 0.f;                                 // (GY2 @ W2^T)[k = lane&31]
#pragma unroll
            for (int c = 0; c < 32; ++c) gh = fmaf(rl(gy, c), w2row[c], gh);
            if (lane < 16) { gpool1 += gh; sY1[j * 16 + lane] = y1; }
            else if (lane < 32) sGA[j * 16 + lane - 16] = gh;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) dwp[k] = fmaf(rl(pool1, k), sgy, dwp[k]);
        __threadfence_block();
        bool found1 = false;
        for (int j = 0; j < x.nrows; ++j) {
            const float y1 = sY1[j * 16 + c1];
            const bool hit = !found1 && y1 == ys1;
            found1 = found1 || hit;
            float ga = sGA[j * 16 + c1] + (hit ? gpool1 : 0.f);
            if (!(fmaf(y1, sc1, sh1) > 0.f)) ga = 0.f;
            if (lane < 16) {
                gz1buf[crow(in, v, V, j, x.has_pad) * 16 + lane] = ga;
                s1 += (double)ga;
                s2 += (double)ga * (double)((y1 - mu1) * is1);
            }
        }
        __threadfence_block();
    }
    float acc[32];
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc[k] = dwp[k]; acc[16 + k] = dwa[k]; }
    block_dw_out<32>(acc, 32, sDW, dw2_parts, w, lane);
    block_stats_out(s1, s2, 16, st1_parts, w, lane);
}

__global__ void __launch_bounds__(256)
k_l1(VfeIn in, const float* __restrict__ W1, const float* __restrict__ bn1, const float* __restrict__ coef1,
     const float* __restrict__ gz1buf, double* __restrict__ dw1_parts) {
    __shared__ float sDW[6 * 16];
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c1 = lane & 15;
    float w1[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w1[k] = W1[k * 16 + c1];
    const float sc1 = bn1[c1], mu1 = bn1[32 + c1], is1 = bn1[48 + c1];
    const float m1 = coef1[c1], m2 = coef1[16 + c1];
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nE = in.ncells - V;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    // pad rows have x = 0 and add nothing to dW1: only real voxels, real rows
    for (int v = blockIdx.x * 4 + w; v < V; v += gridDim.x * 4) {
        Vox x = load_vox(in, v, V, nE);
        for (int t = 0; t < x.s; ++t) {
            float xk[6];
            float y1 = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) { xk[k] = rl(x.xr[k], t); y1 = fmaf(xk[k], w1[k], y1); }
            const int j = x.has_pad ? t + 1 : t;
            const float gz = gz1buf[crow(in, v, V, j, x.has_pad) * 16 + c1];
            const float gy = sc1 * (gz - m1 - (y1 - mu1) * is1 * m2);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] = fmaf(xk[k], gy, acc[k]);
        }
    }
    block_dw_out<6>(acc, 16, sDW, dw1_parts, w, lane);
}

}  // namespace
}  // namespace lisec

using namespace lisec;

namespace {
struct BwdWs {
    float *dout, *gz2, *gz1, *coef;
    double *parts_a, *parts_dw;
    size_t bytes;
    BwdWs(void* base, int cap, int n_points) {
        Carver c(base);
        size_t crows = (size_t)n_points + cap + 2;
        dout = c.take<float>((size_t)(cap + 1) * 64);
        gz2 = c.take<float>(crows * 32);
        gz1 = c.take<float>(crows * 16);
        coef = c.take<float>(2 * 64);
        parts_a = c.take<double>((size_t)kBwdBlocks * 4 * 2 * 64);
        parts_dw = c.take<double>((size_t)kBwdBlocks * 64 * 64);
        bytes = c.off;
    }
};
}  // namespace

extern "C" size_t lisec_vfe_backward_workspace_bytes(int cap_voxels, int n_points) {
    if (cap_voxels < 0 || n_points < 0) return 0;
    return BwdWs(nullptr, cap_voxels, n_points).bytes;
}

extern "C" int lisec_vfe_backward(const lisec_vfe_params* p, const int32_t* info, const int32_t* cell_voxel,
                                  const int32_t* npts, const int32_t* row_start, const float* rows, int n_points,
                                  int ncells, int T, int cap_voxels, const float* saved, const float* dgrid,
                                  float* dout_rows, const float* g_all, const lisec_vfe_grads* g, void* workspace,
                                  size_t workspace_bytes, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(p && info && cell_voxel && npts && row_start && rows && saved && g && workspace, "NULL pointer");
    LISEC_CHECK_ARG((dgrid != nullptr) != (dout_rows != nullptr), "pass either dgrid or dout_rows");
    LISEC_CHECK_ARG(!dout_rows || g_all, "dout_rows needs g_all");
    LISEC_CHECK_ARG(ncells > 0 && T >= 1 && T <= 64 && cap_voxels >= 0 && n_points >= 0, "bad sizes");
    for (int i = 0; i < 3; ++i)
        LISEC_CHECK_ARG(p->kernel[i] && g->kernel[i] && g->gamma[i] && g->beta[i], "NULL VFE parameter/gradient pointer");
    BwdWs ws(workspace, cap_voxels, n_points);
    if (workspace_bytes < ws.bytes) {
        set_error("vfe backward workspace too small: %zu < %zu", workspace_bytes, ws.bytes);
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    VfeSaved sv(const_cast<float*>(saved), cap_voxels);
    VfeIn in{info, npts, row_start, rows, ncells, T, cap_voxels};
    const double N = (double)ncells * (double)T;
    // 1. route the grid gradient to voxels
    if (dgrid) {
        const int gblocks = 1024;
        hipLaunchKernelGGL(k_gather, dim3(gblocks), dim3(256), 0, st, info, cell_voxel, ncells, cap_voxels, dgrid,
                           ws.dout, ws.parts_a);
        hipLaunchKernelGGL(k_virtual_dout, dim3(1), dim3(1024), 0, st, info, cap_voxels, ws.parts_a, gblocks, ws.dout);
    } else {
        // rows [0,V) were computed at the occupied cells only; the virtual voxel gets total - occupied
        ws.dout = dout_rows;
        const int cb = 256;
        hipLaunchKernelGGL(k_rows_colsum, dim3(cb), dim3(256), 0, st, info, cap_voxels, dout_rows, ws.parts_a);
        hipLaunchKernelGGL(k_virtual_from_total, dim3(1), dim3(1024), 0, st, info, cap_voxels, ws.parts_a, cb, g_all,
                           dout_rows);
    }
    LISEC_LAUNCH_CHECK();
    // 2. layer 3 (fcn)
    hipLaunchKernelGGL(k_l3_stats, dim3(kBwdBlocks), dim3(256), 0, st, in, sv.bn3, sv.ymm3, ws.dout, ws.parts_a);
    LISEC_LAUNCH_CHECK();
    if (int rc = launch_bn_bwd_finalize(ws.parts_a, kBwdBlocks, 64, N, g->gamma[2], g->beta[2], ws.coef, st)) return rc;
    size_t lds3 = (size_t)(4 * 2 * kMaxRows * 32 + 64 * 64) * sizeof(float);
    hipLaunchKernelGGL(k_l3, dim3(kBwdBlocks), dim3(256), lds3, st, in, p->kernel[0], p->kernel[1], p->kernel[2],
                       sv.bn1, sv.bn2, sv.bn3, ws.coef, sv.ymm1, sv.ymm2, sv.ymm3, ws.dout, ws.gz2, ws.parts_dw,
                       ws.parts_a);
    LISEC_LAUNCH_CHECK();
    if (int rc = launch_reduce_parts(ws.parts_dw, kBwdBlocks, 64 * 64, 1.0, g->kernel[2], nullptr, st)) return rc;
    if (int rc = launch_bn_bwd_finalize(ws.parts_a, kBwdBlocks, 32, N, g->gamma[1], g->beta[1], ws.coef, st)) return rc;
    // 3. layer 2
    size_t lds2 = (size_t)(4 * 2 * kMaxRows * 16 + 32 * 32) * sizeof(float);
    hipLaunchKernelGGL(k_l2, dim3(kBwdBlocks), dim3(256), lds2, st, in, p->kernel[0], p->kernel[1], sv.bn1, sv.bn2,
                       ws.coef, sv.ymm1, ws.gz2, ws.gz1, ws.parts_dw, ws.parts_a);
    LISEC_LAUNCH_CHECK();
    if (int rc = launch_reduce_parts(ws.parts_dw, kBwdBlocks, 32 * 32, 1.0, g->kernel[1], nullptr, st)) return rc;
    if (int rc = launch_bn_bwd_finalize(ws.parts_a, kBwdBlocks, 16, N, g->gamma[0], g->beta[0], ws.coef, st)) return rc;
    // 4. layer 1
    hipLaunchKernelGGL(k_l1, dim3(kBwdBlocks), dim3(256), 0, st, in, p->kernel[0], sv.bn1, ws.coef, ws.gz1,
                       ws.parts_dw);
    LISEC_LAUNCH_CHECK();
    return launch_reduce_parts(ws.parts_dw, kBwdBlocks, 6 * 16, 1.0, g->kernel[0], nullptr, st);
}
