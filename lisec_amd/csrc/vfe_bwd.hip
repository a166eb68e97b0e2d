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

#include <cstdio>

namespace lisec {
namespace {

constexpr int kBwdBlocks = 512;
constexpr int kL3Blocks = 256;      // k_l3 alone is faster with half the workgroups (49 -> 42 us at 9 400 voxels: its 64x64 partial per
                                    // workgroup dominates); k_l2 / k_l1 / the statistics pass are faster with 512
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
    // (unrolled: the loads of eight voxels in flight at once -- one round trip per voxel made this 42 us at 84 000 voxels)
#pragma unroll 8
    for (int v = blockIdx.x * 4 + ry; v < V; v += gridDim.x * 4) a += (double)dout[(size_t)v * 64 + c];
    red[ry][c] = a;
    __syncthreads();
    if (ry == 0) parts[(size_t)blockIdx.x * 64 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
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
// vparts (compact form, optional): the fp64 column-sum parts of the occupied rows (k_rows_colsum).  The wave that meets the
// virtual voxel (row V) forms its gradient row itself -- (sum of the grid gradient over ALL cells, g_all) - (sum over the
// occupied cells) -- and stores it for the kernels behind this one: what k_virtual_from_total did as a launch of ONE
// workgroup between two others (7 - 26 us on the serial tail of the step).
__global__ void __launch_bounds__(256)
k_l3_stats(VfeIn in, const float* __restrict__ bn3, const float* __restrict__ ymm3,
           float* __restrict__ dout, double* __restrict__ parts, const double* __restrict__ vparts, int nvparts,
           const float* __restrict__ g_all) {
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nvox = V + (in.ncells - V > 0 ? 1 : 0);
    const float sc = bn3[lane], sh = bn3[64 + lane], mu = bn3[128 + lane], is = bn3[192 + lane];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (int v = blockIdx.x * 4 + w; v < nvox; v += gridDim.x * 4) {
        const float ys = sc >= 0.f ? ymm3[(size_t)v * 128 + lane] : ymm3[(size_t)v * 128 + 64 + lane];
        float dv;
        if (vparts && v == V) {
            double a = 0.0;
            int b = 0;
            for (; b + 8 <= nvparts; b += 8) {               // eight loads in flight, added in index order
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = vparts[(size_t)(b + u) * 64 + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += t[u];
            }
            for (; b < nvparts; ++b) a += vparts[(size_t)b * 64 + lane];
            dv = (float)((double)g_all[lane] - a);
            dout[(size_t)V * 64 + lane] = dv;
        } else {
            dv = dout[(size_t)v * 64 + lane];
        }
        const float gz = fmaf(ys, sc, sh) > 0.f ? dv : 0.f;
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


// ================================================================================================================
// Tiled (MFMA) backward of layers 3 and 2.  The per-voxel kernels above walk a voxel's rows one after the other with
// lane = channel and v_readlane broadcasts: fine for a 20 k-point sweep (2.6 class rows per voxel), 0.6 ms on a
// 200 k-point one.  Here the class rows ("slots": slot of row j of voxel v = row_start[v] + v + j, slot 0 = pad row)
// are cut into 32-slot tiles regardless of voxel borders; per tile and layer three contractions run on the matrix cores
//   y3 = [pool2 | a2] @ W3            (layer 3 only: the layer-2 pre-BN rows come from the forward, `y2rows`)
//   dW += [pool | a]^T @ gy           (64x64 / 32x32 accumulators kept in registers across the wave's tiles)
//   gh  = gy @ W^T                    (gradient wrt the layer's 2C inputs, stored per slot)
// and the max-pool routing needs no search: the forward recorded the slot of the first row holding each per-voxel
// maximum / minimum (`arg`), so "is this the winner row" is an integer compare.  What stays per voxel and sequential
// (k_post: the pooled half of gh summed over the voxel's rows and handed to the winner row one layer down, the ReLU
// gate, the BN-backward statistics) is elementwise.  lane = slot for everything elementwise (lanes l and l+32 hold the
// same slot: an MFMA A operand wants k = 2*step + lane/32), LDS tiles with a 68-float row stride for the transposes.
constexpr int kLdT = 68;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TileCtx {
    const int* slot_vox;      // voxel of every slot
    const float* pool1;       // [nvox][16] pooled output of layer 1 per voxel
    const float* pool2;       // [nvox][32]
    const unsigned char* aw1; // [nvox][16] winner slot of layer 1 (max or min picked by the sign of the BN scale)
    const unsigned char* aw2; // [nvox][32]
    const unsigned char* aw3; // [nvox][64]
};

// per voxel: which slots it owns, its pooled inputs, its winner slots
__global__ void __launch_bounds__(256)
k_slot_prep(VfeIn in, const float* __restrict__ bn1, const float* __restrict__ bn2, const float* __restrict__ bn3,
            const float* __restrict__ ymm1, const float* __restrict__ ymm2, const unsigned char* __restrict__ arg1,
            const unsigned char* __restrict__ arg2, const unsigned char* __restrict__ arg3, int* __restrict__ slot_vox,
            float* __restrict__ pool1, float* __restrict__ pool2, unsigned char* __restrict__ aw1,
            unsigned char* __restrict__ aw2, unsigned char* __restrict__ aw3) {
    const int lane = lane_id(), w = threadIdx.x >> 6;
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nvox = V + (in.ncells - V > 0 ? 1 : 0);
    const int c1 = lane & 15, c2 = lane & 31;
    const float sc1 = bn1[c1], sh1 = bn1[16 + c1], sc2 = bn2[c2], sh2 = bn2[32 + c2], sc3 = bn3[lane];
    for (int v = blockIdx.x * 4 + w; v < nvox; v += gridDim.x * 4) {
        const bool virt = v == V;
        const int s = virt ? 0 : in.npts[v];
        const int base = (virt ? in.info[LISEC_VI_NROWS] : in.row_start[v]) + v;
        if (lane <= s) slot_vox[base + lane] = v;
        if (lane < 16) {
            pool1[(size_t)v * 16 + lane] = pool_from(ymm1[(size_t)v * 32 + lane], ymm1[(size_t)v * 32 + 16 + lane], sc1, sh1);
            aw1[(size_t)v * 16 + lane] = arg1[(size_t)v * 32 + (sc1 >= 0.f ? 0 : 16) + lane];
        }
        if (lane < 32) {
            pool2[(size_t)v * 32 + lane] = pool_from(ymm2[(size_t)v * 64 + lane], ymm2[(size_t)v * 64 + 32 + lane], sc2, sh2);
            aw2[(size_t)v * 32 + lane] = arg2[(size_t)v * 64 + (sc2 >= 0.f ? 0 : 32) + lane];
        }
        aw3[(size_t)v * 64 + lane] = arg3[(size_t)v * 128 + (sc3 >= 0.f ? 0 : 64) + lane];
    }
}

struct SlotInfo {
    int v, j;
    bool valid;
    float wr;
    int row;      // index into in.rows of the slot's point (-1: pad row)
};

__device__ __forceinline__ SlotInfo slot_info(const VfeIn& in, const int* __restrict__ slot_vox, int sl, int S, int V,
                                              int nE) {
    SlotInfo q;
    const bool inb = sl < S;
    q.v = inb ? slot_vox[sl] : 0;
    const bool virt = q.v == V;
    const int s = (!inb || virt) ? 0 : in.npts[q.v];
    const int rs = virt ? in.info[LISEC_VI_NROWS] : in.row_start[q.v];
    q.j = inb ? sl - (rs + q.v) : 0;                 // beyond the last slot (tail of the last tile): no row at all
    const bool has_pad = virt || s < in.T;
    q.valid = inb && (q.j > 0 || has_pad);
    q.wr = q.j == 0 ? (virt ? (float)in.T * (float)nE : (float)(in.T - s)) : 1.f;
    q.row = (inb && q.j > 0) ? rs + q.j - 1 : -1;
    return q;
}

// LAYER 3: C = 64 channels, A' = [pool2 (32) | a2 (32)];  LAYER 2: C = 32, A' = [pool1 (16) | a1 (16)].
// dyn LDS: per wave two tiles [32][kLdT]; then W [2C'][C] and W^T (C' = C input half-width * 2 = C), constants.
template <int LAYER>
__global__ void __launch_bounds__(256, 1)          // one wave per SIMD: the whole 512-register file
k_bwd_tile(VfeIn in, TileCtx cx, const float* __restrict__ Wl /* (C, C) kernel of the layer */,
           const float* __restrict__ W1, const float* __restrict__ bn_lo /* bnstate of the layer below */,
           const float* __restrict__ bn_l /* bnstate of this layer */, const float* __restrict__ coef,
           const float* __restrict__ y2rows, const float* __restrict__ dsrc /* L3: dout[v][64]; L2: gz2[slot][32] */,
           float* __restrict__ ghbuf, double* __restrict__ dw_parts) {
    constexpr int C = LAYER == 3 ? 64 : 32;         // channels of this layer == width of A'
    constexpr int H = C / 2;                        // pooled / pointwise half width
    constexpr int NB = C / 32;                      // 32-column blocks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    float* T1 = smem + w * (2 * 32 * kLdT);          // A' tile (lane = slot rows)
    float* T2 = T1 + 32 * kLdT;                      // y3 / gy tile
    float* sW = smem + 4 * (2 * 32 * kLdT);          // W[k][c]
    float* sWT = sW + C * C;                         // W^T[c][k]
    float* sK = sWT + C * C;                         // constants: scale, shift, mean, invstd, m1, m2 of this layer [6][C]
    float* sLo = sK + 6 * C;                         // scale, shift of the layer below [2][H]
    float* sW1 = sLo + 2 * H;                        // W1 (6 x 16), layer 2 only
    for (int i = threadIdx.x; i < C * C; i += 256) {
        const float v = Wl[i];
        sW[i] = v;
        sWT[(i % C) * C + i / C] = v;
    }
    for (int i = threadIdx.x; i < 4 * C; i += 256) sK[i] = bn_l[i];
    for (int i = threadIdx.x; i < 2 * C; i += 256) sK[4 * C + i] = coef[i];
    for (int i = threadIdx.x; i < 2 * H; i += 256) sLo[i] = bn_lo[i];
    if (LAYER == 2) for (int i = threadIdx.x; i < 96; i += 256) sW1[i] = W1[i];
    __syncthreads();
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nE = in.ncells - V;
    const int S = in.info[LISEC_VI_NROWS] + V + (nE > 0 ? 1 : 0);
    const int ntiles = (S + 31) / 32;
    f32x16 dW[NB][NB];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) dW[a][b] = {0};
    for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += gridDim.x * 4) {
        const int sl = tile * 32 + l31;
        const SlotInfo q = slot_info(in, cx.slot_vox, sl, S, V, nE);
        // Lanes l and l + 32 hold the same slot; each OWNS the channels 2*i + half (what an MFMA A operand wants of it).
        // ---- A' = [pooled inputs of the voxel | this slot's activation of the layer below], own half: Ah[i] = A'[2i+half]
        float Ah[C / 2];
        {
            const float4* pp = reinterpret_cast<const float4*>((LAYER == 3 ? cx.pool2 : cx.pool1) + (size_t)q.v * H);
#pragma unroll
            for (int i = 0; i < H / 4; ++i) {
                const float4 t = pp[i];
                Ah[2 * i] = half ? t.y : t.x;
                Ah[2 * i + 1] = half ? t.w : t.z;
            }
        }
        float y2h[16];                                  // layer-2 pre-BN row of the slot, own half
        {
            const float4* yp = reinterpret_cast<const float4*>(y2rows + (size_t)(sl < S ? sl : 0) * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 t = yp[i];
                y2h[2 * i] = half ? t.y : t.x;
                y2h[2 * i + 1] = half ? t.w : t.z;
            }
        }
        if (LAYER == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Ah[H / 2 + i] = bnrelu(y2h[i], sLo[2 * i + half], sLo[H + 2 * i + half]);
        } else {
            float x[6] = {0, 0, 0, 0, 0, 0};
            if (q.row >= 0) {
                const float2* r = reinterpret_cast<const float2*>(in.rows + (size_t)q.row * 6);
                const float2 p0 = r[0], p1 = r[1], p2 = r[2];
                x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 2 * i + half;
                float y1 = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) y1 = fmaf(x[k], sW1[k * 16 + c], y1);
                Ah[H / 2 + i] = bnrelu(y1, sLo[c], sLo[H + c]);
            }
        }
        if (!q.valid) {                                 // unused pad slot of a full voxel / beyond the last slot: its
#pragma unroll                                          // saved rows were never written
            for (int i = 0; i < C / 2; ++i) Ah[i] = 0.f;
        }
        // ---- gy, own half: gyh[i] = gy[2i + half] ----------------------------------------------------------------
        float gyh[C / 2];
        if (LAYER == 3) {
            f32x16 acc[2] = {{0}, {0}};
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float* wr_ = sW + (2 * kk + half) * 64 + l31;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ah[kk], wr_[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ah[kk], wr_[32], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                T2[row * kLdT + l31] = acc[0][r];
                T2[row * kLdT + 32 + l31] = acc[1][r];
            }
            __threadfence_block();
            const unsigned char* ap = cx.aw3 + (size_t)q.v * 64;
            const float* dp = dsrc + (size_t)q.v * 64;
#pragma unroll
            for (int c4 = 0; c4 < 16; ++c4) {
                const float4 d4 = *reinterpret_cast<const float4*>(dp + 4 * c4);
                const uchar4 a4 = *reinterpret_cast<const uchar4*>(ap + 4 * c4);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = 2 * c4 + u, c = 2 * i + half;          // c = 4*c4 + 2*u + half
                    const float yv = T2[l31 * kLdT + c];
                    const float dv = u ? (half ? d4.w : d4.z) : (half ? d4.y : d4.x);
                    const int av = u ? (half ? a4.w : a4.z) : (half ? a4.y : a4.x);
                    const float sc = sK[c], sh = sK[C + c], mu = sK[2 * C + c], is = sK[3 * C + c];
                    const float m1 = sK[4 * C + c], m2 = sK[5 * C + c];
                    const float gz = (av == q.j && fmaf(yv, sc, sh) > 0.f) ? dv : 0.f;
                    const float yh = (yv - mu) * is;
                    gyh[i] = q.valid ? sc * (gz - q.wr * m1 - q.wr * yh * m2) : 0.f;
                }
            }
            __threadfence_block();                      // T2 is rewritten below
        } else {
            const float* gp = dsrc + (size_t)(sl < S ? sl : 0) * 32;
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gp + 4 * c4);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = 2 * c4 + u, c = 2 * i + half;
                    const float gv = u ? (half ? g4.w : g4.z) : (half ? g4.y : g4.x);
                    const float sc = sK[c], mu = sK[2 * C + c], is = sK[3 * C + c];
                    const float m1 = sK[4 * C + c], m2 = sK[5 * C + c];
                    const float yh = (y2h[i] - mu) * is;
                    gyh[i] = q.valid ? sc * (gv - q.wr * m1 - q.wr * yh * m2) : 0.f;
                }
            }
        }
        // ---- dW += A'^T gy : operands through LDS ([slot][channel] tiles), every lane stores the channels it owns
#pragma unroll
        for (int i = 0; i < C / 2; ++i) {
            T1[l31 * kLdT + 2 * i + half] = Ah[i];
            T2[l31 * kLdT + 2 * i + half] = gyh[i];
        }
        __threadfence_block();
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const float* ar = T1 + (2 * s2 + half) * kLdT + l31;
            const float* br = T2 + (2 * s2 + half) * kLdT + l31;
#pragma unroll
            for (int mb = 0; mb < NB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    dW[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[mb * 32], br[nb * 32], dW[mb][nb], 0, 0, 0);
        }
        // ---- gh = gy W^T, stored per slot ---------------------------------------------------------------------
        f32x16 gh[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) gh[nb] = {0};
#pragma unroll
        for (int cc = 0; cc < C / 2; ++cc) {
            const float* wt = sWT + (2 * cc + half) * C + l31;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                gh[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(gyh[cc], wt[nb * 32], gh[nb], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int slot = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (slot < S) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) ghbuf[(size_t)slot * C + nb * 32 + l31] = gh[nb][r];
            }
        }
        __threadfence_block();                          // the tiles are reused by the wave's next tile
    }
    // ---- dW of the four waves, added in wave order, as per-workgroup fp64 partials [k][c] -----------------------
    float* sDW = smem;                                   // C x C floats (the tile area is free now)
    for (int round = 0; round < 4; ++round) {
        __syncthreads();
        if (w == round) {
#pragma unroll
            for (int mb = 0; mb < NB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, c = nb * 32 + l31;
                        sDW[k * C + c] = (round == 0 ? 0.f : sDW[k * C + c]) + dW[mb][nb][r];
                    }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) dw_parts[(size_t)blockIdx.x * C * C + i] = (double)sDW[i];
}

// Per voxel, rows in order (lane = channel of the layer BELOW, C2 = 32 for LAYER 3, 16 for LAYER 2): the pooled half
// of gh summed over the voxel's rows goes to that layer's winner row, the ReLU gate, the gradient wrt the pre-BN value
// of the layer below per slot, and its BatchNormalization-backward statistics.
template <int LAYER>
__global__ void __launch_bounds__(256)
k_post(VfeIn in, const float* __restrict__ W1, const float* __restrict__ bn_lo, const unsigned char* __restrict__ aw_lo,
       const float* __restrict__ y2rows, const float* __restrict__ ghbuf, float* __restrict__ gzbuf,
       double* __restrict__ st_parts) {
    constexpr int C = LAYER == 3 ? 64 : 32, C2 = C / 2;
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & (C2 - 1);
    int V = in.info[LISEC_VI_NVOX];
    if (V > in.cap) V = in.cap;
    const int nE = in.ncells - V;
    const int nvox = V + (nE > 0 ? 1 : 0);
    const float sc = bn_lo[c], sh = bn_lo[C2 + c], mu = bn_lo[2 * C2 + c], is = bn_lo[3 * C2 + c];
    float w1[6] = {0, 0, 0, 0, 0, 0};
    if (LAYER == 2) {
#pragma unroll
        for (int k = 0; k < 6; ++k) w1[k] = W1[k * 16 + c];
    }
    double s1 = 0.0, s2 = 0.0;
    for (int v = blockIdx.x * 4 + w; v < nvox; v += gridDim.x * 4) {
        Vox x = load_vox(in, v, V, nE);
        const size_t base = (size_t)(v == V ? in.info[LISEC_VI_NROWS] : x.rs) + v;
        const int j0 = x.has_pad ? 0 : 1;
        const int win = aw_lo[(size_t)v * C2 + c];
        float gpool = 0.f;
        for (int j = j0; j <= x.s; ++j) gpool += ghbuf[(base + j) * C + c];
        for (int j = j0; j <= x.s; ++j) {
            float y;
            if (LAYER == 3) {
                y = y2rows[(base + j) * 32 + c];
            } else {
                y = 0.f;
                if (j > 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) y = fmaf(rl(x.xr[k], j - 1), w1[k], y);
                }
            }
            float ga = ghbuf[(base + j) * C + C2 + c] + (win == j ? gpool : 0.f);
            if (!(fmaf(y, sc, sh) > 0.f)) ga = 0.f;
            if (lane < C2) {
                gzbuf[(base + j) * C2 + lane] = ga;
                s1 += (double)ga;
                s2 += (double)ga * (double)((y - mu) * is);
            }
        }
    }
    block_stats_out(s1, s2, C2, st_parts, w, lane);
}

}  // namespace
}  // namespace lisec

using namespace lisec;

namespace {
struct BwdWs {
    float *dout, *gz2, *gz1, *coef;
    double *parts_a, *parts_dw;
    // tiled path
    int* slot_vox;
    float *pool1, *pool2, *gh;
    unsigned char *aw1, *aw2, *aw3;
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
        slot_vox = c.take<int>(crows + 32);
        pool1 = c.take<float>((size_t)(cap + 1) * 16);
        pool2 = c.take<float>((size_t)(cap + 1) * 32);
        gh = c.take<float>((crows + 32) * 64);
        aw1 = c.take<unsigned char>((size_t)(cap + 1) * 16);
        aw2 = c.take<unsigned char>((size_t)(cap + 1) * 32);
        aw3 = c.take<unsigned char>((size_t)(cap + 1) * 64);
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
                                  float* dout_rows, const float* g_all, const lisec_vfe_grads* g, int flags,
                                  void* workspace, size_t workspace_bytes, lisec_stream_t stream_) {
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
    const bool tiled = (flags & LISEC_VFE_BWD_TILED) != 0;
    VfeSaved sv(const_cast<float*>(saved), cap_voxels, tiled ? n_points : 0);
    LISEC_CHECK_ARG(!tiled || n_points > 0, "the tiled backward needs the per-row extras (n_points > 0)");
    VfeIn in{info, npts, row_start, rows, ncells, T, cap_voxels};
    const double N = (double)ncells * (double)T;
    // 1. route the grid gradient to voxels
    const double* vparts = nullptr;
    int vparts_n = 0;
    if (dgrid) {
        const int gblocks = 1024;
        LISEC_LAUNCH(k_gather, dim3(gblocks), dim3(256), 0, st, info, cell_voxel, ncells, cap_voxels, dgrid,
                           ws.dout, ws.parts_a);
        LISEC_LAUNCH(k_virtual_dout, dim3(1), dim3(1024), 0, st, info, cap_voxels, ws.parts_a, gblocks, ws.dout);
    } else {
        // rows [0,V) were computed at the occupied cells only; the virtual voxel gets total - occupied
        // (the parts go where the tile kernels put their weight-gradient partials later; k_l3_stats writes parts_a itself)
        ws.dout = dout_rows;
        vparts_n = cap_voxels >= 32768 ? 256 : (cap_voxels >= 4096 ? 64 : 16);
        vparts = ws.parts_dw;
        LISEC_LAUNCH(k_rows_colsum, dim3(vparts_n), dim3(256), 0, st, info, cap_voxels, dout_rows, ws.parts_dw);
    }
    LISEC_LAUNCH_CHECK();
    // 2. layer 3 (fcn)
    LISEC_LAUNCH(k_l3_stats, dim3(kBwdBlocks), dim3(256), 0, st, in, sv.bn3, sv.ymm3, ws.dout, ws.parts_a, vparts, vparts_n,
                 g_all);
    LISEC_LAUNCH_CHECK();
    if (int rc = launch_bn_bwd_finalize(ws.parts_a, kBwdBlocks, 64, N, g->gamma[2], g->beta[2], ws.coef, st)) return rc;
    // lisec_tuning.debug_sync (diagnostic): synchronise and report after every launch of this call
    const bool dbg = tuning().debug_sync != 0;
#define LISEC_DBG(WHAT_)                                                                       \
    do {                                                                                       \
        if (dbg) {                                                                             \
            hipError_t e_ = hipStreamSynchronize(st);                                          \
            fprintf(stderr, "[lisec_vfe_backward] %s: %s\n", WHAT_, hipGetErrorString(e_));    \
            if (e_ != hipSuccess) { set_error("%s failed", WHAT_); return LISEC_EHIP; }        \
        }                                                                                      \
    } while (0)
    LISEC_DBG("statistics of layer 3");
    if (tiled) {
        // layers 3 and 2 on 32-slot tiles (matrix cores) + the per-voxel elementwise passes
        constexpr int kTileBlocks = 256;                 // one 4-wave workgroup per CU (104 KB of LDS for layer 3)
        // the per-voxel passes are chains of dependent loads: a wave per voxel and enough waves to hide them (the
        // statistic partials of k_post cap the grid: parts_a holds 2048 rows of 2 x 32 doubles)
        int vblocks = cdiv((long long)cap_voxels + 1, 4);
        if (vblocks > 2048) vblocks = 2048;
        if (vblocks < 1) vblocks = 1;
        TileCtx cx{ws.slot_vox, ws.pool1, ws.pool2, ws.aw1, ws.aw2, ws.aw3};
        LISEC_LAUNCH(k_slot_prep, dim3(vblocks), dim3(256), 0, st, in, sv.bn1, sv.bn2, sv.bn3, sv.ymm1, sv.ymm2,
                           sv.arg1, sv.arg2, sv.arg3, ws.slot_vox, ws.pool1, ws.pool2, ws.aw1, ws.aw2, ws.aw3);
        LISEC_DBG("k_slot_prep");
        const size_t ldsT3 = (size_t)(4 * 2 * 32 * kLdT + 2 * 64 * 64 + 6 * 64 + 2 * 32 + 96) * sizeof(float);
        LISEC_LAUNCH(k_bwd_tile<3>, dim3(kTileBlocks), dim3(256), ldsT3, st, in, cx, p->kernel[2], p->kernel[0],
                           sv.bn2, sv.bn3, ws.coef, sv.y2rows, ws.dout, ws.gh, ws.parts_dw);
        LISEC_LAUNCH_CHECK();
        LISEC_DBG("k_bwd_tile<3>");
        LISEC_LAUNCH(k_post<3>, dim3(vblocks), dim3(256), 0, st, in, p->kernel[0], sv.bn2, ws.aw2, sv.y2rows,
                           ws.gh, ws.gz2, ws.parts_a);
        LISEC_LAUNCH_CHECK();
        LISEC_DBG("k_post<3>");
        // (the tile kernel's weight-gradient partials and k_post's BatchNormalization partials: one launch)
        if (int rc = launch_reduce_and_bwd_finalize(ws.parts_dw, kTileBlocks, 64 * 64, 1.0, g->kernel[2], ws.parts_a, vblocks, 32,
                                                    N, g->gamma[1], g->beta[1], ws.coef, st)) return rc;
        const size_t ldsT2 = (size_t)(4 * 2 * 32 * kLdT + 2 * 32 * 32 + 6 * 32 + 2 * 16 + 96) * sizeof(float);
        LISEC_LAUNCH(k_bwd_tile<2>, dim3(kTileBlocks), dim3(256), ldsT2, st, in, cx, p->kernel[1], p->kernel[0],
                           sv.bn1, sv.bn2, ws.coef, sv.y2rows, ws.gz2, ws.gh, ws.parts_dw);
        LISEC_LAUNCH_CHECK();
        LISEC_DBG("k_bwd_tile<2>");
        LISEC_LAUNCH(k_post<2>, dim3(vblocks), dim3(256), 0, st, in, p->kernel[0], sv.bn1, ws.aw1, sv.y2rows,
                           ws.gh, ws.gz1, ws.parts_a);
        LISEC_LAUNCH_CHECK();
        LISEC_DBG("k_post<2>");
        if (int rc = launch_reduce_and_bwd_finalize(ws.parts_dw, kTileBlocks, 32 * 32, 1.0, g->kernel[1], ws.parts_a, vblocks, 16,
                                                    N, g->gamma[0], g->beta[0], ws.coef, st)) return rc;
    } else {
        size_t lds3 = (size_t)(4 * 2 * kMaxRows * 32 + 64 * 64) * sizeof(float);
        LISEC_LAUNCH(k_l3, dim3(kL3Blocks), dim3(256), lds3, st, in, p->kernel[0], p->kernel[1], p->kernel[2],
                           sv.bn1, sv.bn2, sv.bn3, ws.coef, sv.ymm1, sv.ymm2, sv.ymm3, ws.dout, ws.gz2, ws.parts_dw,
                           ws.parts_a);
        LISEC_LAUNCH_CHECK();
        // (weight-gradient partial sums and the BatchNormalization finaliser of the same producer: one launch)
        if (int rc = launch_reduce_and_bwd_finalize(ws.parts_dw, kL3Blocks, 64 * 64, 1.0, g->kernel[2], ws.parts_a, kL3Blocks,
                                                    32, N, g->gamma[1], g->beta[1], ws.coef, st)) return rc;
        // 3. layer 2
        size_t lds2 = (size_t)(4 * 2 * kMaxRows * 16 + 32 * 32) * sizeof(float);
        LISEC_LAUNCH(k_l2, dim3(kBwdBlocks), dim3(256), lds2, st, in, p->kernel[0], p->kernel[1], sv.bn1, sv.bn2,
                           ws.coef, sv.ymm1, ws.gz2, ws.gz1, ws.parts_dw, ws.parts_a);
        LISEC_LAUNCH_CHECK();
        if (int rc = launch_reduce_and_bwd_finalize(ws.parts_dw, kBwdBlocks, 32 * 32, 1.0, g->kernel[1], ws.parts_a, kBwdBlocks,
                                                    16, N, g->gamma[0], g->beta[0], ws.coef, st)) return rc;
    }
    // 4. layer 1
    LISEC_LAUNCH(k_l1, dim3(kBwdBlocks), dim3(256), 0, st, in, p->kernel[0], sv.bn1, ws.coef, ws.gz1,
                       ws.parts_dw);
    LISEC_LAUNCH_CHECK();
    return launch_reduce_parts(ws.parts_dw, kBwdBlocks, 6 * 16, 1.0, g->kernel[0], nullptr, st);
}
