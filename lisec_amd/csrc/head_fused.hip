// Upsampling branches + heads collapsed into 16-channel contractions (exact, by linearity).
//
// createModel ends with (model_training.py:246-255)
//     up_b = Conv2DTranspose(256, k_b, s_b, 'same')(x_b)          b = 1..3: bias, NO BatchNormalization, NO activation
//     cat  = Concatenate()([up_1, up_2, up_3])                    (100, 200, 768)
//     cls, reg = Conv2D(2, 1)(cat), Conv2D(14, 1)(cat)            linear 1x1 heads, H = [W_cls | W_reg] (768, 16)
// so   head[m, j] = b_j + sum_b sum_n up_b[m, n] H[256 b + n, j]
//                 = b'_j + sum_b sum_{tap, c} x_b[src_b(m, tap), c] W'_b[tap][c][j]
// with the COMPOSITE kernels  W'_b[tap][c][j] = sum_n W_b[tap][n][c] H[256 b + n][j]   (Keras layout W_b: (kh, kw, out, in))
// and  b'_j = b_j + sum_b sum_n bias_b[n] H[256 b + n][j]:  three transposed convolutions to 16 channels instead of three to
// 256 channels and a 768 -> 16 contraction -- 1 GFLOP instead of 16.2 forward, and the (100, 200, 768) concat tensor and its
// gradient (61 MB each) are never formed.  The gradients follow from the same associativity: with G_b = dL/dW'_b (the
// weight gradient of the 16-channel contraction, (taps, Cin, 16)) and S_j = sum_m dhead[m, j]
//     dW_b[tap][n][c] = sum_j G_b[tap][c][j] H[256 b + n][j]
//     dH[256 b + n][j] = sum_{tap, c} W_b[tap][n][c] G_b[tap][c][j] + bias_b[n] S_j      (up_b carries its bias into the heads)
//     dbias_b[n]      = sum_j S_j H[256 b + n][j]                    db_j = S_j
// and dL/dx_b is the data gradient of the 16-channel contraction.  Every sum below runs in a fixed order (deterministic).
#include "common.h"

namespace lisec {
namespace {

constexpr int kHeadCols = 16;
constexpr int kMaxUp = 512;                      // channels of an upsampling branch (the reference: 256)

// Workgroup = 16 consecutive (tap, c) pairs x 16 slices of n: thread (pair p, slice q) sums n = q, q + 16, ... and the
// slices are added in slice order through LDS (fixed order: deterministic).  A thread per (tap, c) walking all Cup rows
// (5 workgroups for the first branch) took 60 us at the head of every step, in front of the weight repack.
constexpr int kPairs = 16, kSlices = 16;
__global__ void __launch_bounds__(256)
k_head_compose(const float* __restrict__ up_kernel, const float* __restrict__ up_bias, const float* __restrict__ head_w,
               int taps, int Cin, int Cup, long long ts, long long cs, float* __restrict__ Wc,
               const float* __restrict__ bias_in, float* __restrict__ bias_out) {
    __shared__ float sH[kMaxUp * kHeadCols];
    __shared__ float red[kSlices][kPairs][kHeadCols + 1];
    for (int i = threadIdx.x; i < Cup * kHeadCols; i += 256) sH[i] = head_w[i];
    __syncthreads();
    const int pl = threadIdx.x & (kPairs - 1), q = threadIdx.x / kPairs;
    const int idx = blockIdx.x * kPairs + pl;
    const bool live = idx < taps * Cin;
    const int tap = live ? idx / Cin : 0, c = live ? idx - tap * Cin : 0;
    float acc[kHeadCols];
#pragma unroll
    for (int j = 0; j < kHeadCols; ++j) acc[j] = 0.f;
    const float* w = up_kernel + (size_t)tap * Cup * Cin + c;
    for (int n = q; n < Cup; n += kSlices) {
        const float v = live ? w[(size_t)n * Cin] : 0.f;
#pragma unroll
        for (int j = 0; j < kHeadCols; ++j) acc[j] = fmaf(v, sH[n * kHeadCols + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < kHeadCols; ++j) red[q][pl][j] = acc[j];
    __syncthreads();
    {
        const int p2 = threadIdx.x / kHeadCols, j = threadIdx.x % kHeadCols;      // 16 pairs x 16 columns
        const int id2 = blockIdx.x * kPairs + p2;
        if (id2 < taps * Cin) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < kSlices; ++k) v += red[k][p2][j];
            const int t2 = id2 / Cin, c2 = id2 - t2 * Cin;
            Wc[t2 * ts + c2 * cs + j] = v;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < kHeadCols && bias_out) {
        const int j = threadIdx.x;
        float b = bias_in ? bias_in[j] : 0.f;
        for (int n = 0; n < Cup; ++n) b = fmaf(up_bias[n], sH[n * kHeadCols + j], b);
        bias_out[j] = b;
    }
}

// dW_b[tap][n][c] = sum_j G[tap][c][j] H[n][j]:  thread (tap, c) of 64 consecutive pairs x one of 4 n-quarters of the
// 32 rows blockIdx.y owns
__global__ void __launch_bounds__(256)
k_head_compose_bwd_w(const float* __restrict__ G, long long ts, long long cs, const float* __restrict__ head_w, int taps,
                     int Cin, int Cup, float* __restrict__ d_up_kernel) {
    __shared__ float sH[32 * kHeadCols];
    const int n0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 32 * kHeadCols; i += 256) sH[i] = n0 * kHeadCols + i < Cup * kHeadCols ? head_w[n0 * kHeadCols + i] : 0.f;
    __syncthreads();
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63), nq = threadIdx.x >> 6;
    if (idx >= taps * Cin) return;
    const int tap = idx / Cin, c = idx - tap * Cin;
    float gq[kHeadCols];
    const float* gp = G + tap * ts + c * cs;
#pragma unroll
    for (int j = 0; j < kHeadCols; ++j) gq[j] = gp[j];
    float* o = d_up_kernel + (size_t)tap * Cup * Cin + c;
    for (int i = nq * 8; i < nq * 8 + 8; ++i) {
        const int n = n0 + i;
        if (n >= Cup) break;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < kHeadCols; ++j) v = fmaf(gq[j], sH[i * kHeadCols + j], v);
        o[(size_t)n * Cin] = v;
    }
}

// one workgroup per n:  dH[n][j] = sum_{tap, c} W_b[tap][n][c] G[tap][c][j] + bias_b[n] S_j;  dbias_b[n] = sum_j S_j H[n][j]
__global__ void __launch_bounds__(256)
k_head_compose_bwd_h(const float* __restrict__ G, long long ts, long long cs, const float* __restrict__ up_kernel,
                     const float* __restrict__ up_bias, const float* __restrict__ head_w, const float* __restrict__ S,
                     int taps, int Cin, int Cup, float* __restrict__ d_head_w, float* __restrict__ d_up_bias) {
    __shared__ float red[256][kHeadCols + 1];
    const int n = blockIdx.x;
    float acc[kHeadCols];
#pragma unroll
    for (int j = 0; j < kHeadCols; ++j) acc[j] = 0.f;
    const int total = taps * Cin;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int tap = idx / Cin, c = idx - tap * Cin;
        const float w = up_kernel[((size_t)tap * Cup + n) * Cin + c];
        const float* gp = G + tap * ts + c * cs;
#pragma unroll
        for (int j = 0; j < kHeadCols; ++j) acc[j] = fmaf(w, gp[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < kHeadCols; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {         // fixed tree: the same order every run
        if (threadIdx.x < half) {
#pragma unroll
            for (int j = 0; j < kHeadCols; ++j) red[threadIdx.x][j] += red[threadIdx.x + half][j];
        }
        __syncthreads();
    }
    if (threadIdx.x < kHeadCols)
        d_head_w[(size_t)n * kHeadCols + threadIdx.x] = fmaf(up_bias ? up_bias[n] : 0.f, S[threadIdx.x], red[0][threadIdx.x]);
    if (threadIdx.x == 0 && d_up_bias) {
        float b = 0.f;
        for (int j = 0; j < kHeadCols; ++j) b = fmaf(S[j], head_w[(size_t)n * kHeadCols + j], b);
        d_up_bias[n] = b;
    }
}

// The kernel == stride branches run as 1x1 contractions whose columns are (tap, j): T[pos][tap * 16 + j], pos = (h / ps,
// w / ps), tap = (h % ps) * ps + w % ps.   forward: head[m][j] += sum over branches of T[...];   backward: dT[...] = dhead[m][j]
struct ShuffleBranches { int n; int ps[4]; float* T[4]; };
template <bool BACKWARD>
__global__ void __launch_bounds__(256)
k_head_shuffle(float* __restrict__ head, int Ho, int Wo, ShuffleBranches br) {
    const int i4 = blockIdx.x * 256 + threadIdx.x;                   // one float4 of the (M, 16) head map
    if (i4 >= Ho * Wo * (kHeadCols / 4)) return;
    const int m = i4 >> 2, q = i4 & 3;
    const int h = m / Wo, w = m - h * Wo;
    float4 v = *reinterpret_cast<const float4*>(head + (size_t)i4 * 4);
    for (int b = 0; b < br.n; ++b) {
        const int ps = br.ps[b];
        const int pos = (h / ps) * (Wo / ps) + w / ps, tap = (h % ps) * ps + w % ps;
        float4* t = reinterpret_cast<float4*>(br.T[b] + ((size_t)pos * ps * ps + tap) * kHeadCols) + q;
        if (BACKWARD) *t = v;
        else { const float4 u = *t; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    }
    if (!BACKWARD) *reinterpret_cast<float4*>(head + (size_t)i4 * 4) = v;
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" int lisec_head_compose(const float* up_kernel, const float* up_bias, const float* head_w, int taps, int Cin,
                                  int Cup, long long out_tap_stride, long long out_c_stride, float* Wc,
                                  const float* bias_in, float* bias_out, lisec_stream_t stream) {
    LISEC_CHECK_ARG(up_kernel && head_w && Wc && taps > 0 && Cin > 0 && Cup > 0 && Cup <= kMaxUp, "bad compose arguments");
    LISEC_CHECK_ARG(!bias_out || up_bias, "a composite bias needs the branch bias");
    LISEC_LAUNCH(k_head_compose, dim3(cdiv((long long)taps * Cin, kPairs)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       up_kernel, up_bias, head_w, taps, Cin, Cup, out_tap_stride, out_c_stride, Wc, bias_in, bias_out);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_head_compose_backward(const float* G, long long g_tap_stride, long long g_c_stride,
                                           const float* up_kernel, const float* up_bias, const float* head_w, const float* S,
                                           int taps, int Cin, int Cup, float* d_up_kernel, float* d_up_bias, float* d_head_w,
                                           lisec_stream_t stream) {
    LISEC_CHECK_ARG(G && up_kernel && head_w && S && d_up_kernel && d_head_w && taps > 0 && Cin > 0 && Cup > 0 && Cup <= kMaxUp,
                    "bad compose-backward arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    LISEC_LAUNCH(k_head_compose_bwd_w, dim3(cdiv((long long)taps * Cin, 64), cdiv(Cup, 32)), dim3(256), 0, st, G,
                       g_tap_stride, g_c_stride, head_w, taps, Cin, Cup, d_up_kernel);
    LISEC_LAUNCH(k_head_compose_bwd_h, dim3(Cup), dim3(256), 0, st, G, g_tap_stride, g_c_stride, up_kernel, up_bias, head_w, S,
                       taps, Cin, Cup, d_head_w, d_up_bias);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_head_shuffle(float* head, int Ho, int Wo, int n_branches, float* const* T, const int* ps, int backward,
                                  lisec_stream_t stream) {
    LISEC_CHECK_ARG(head && Ho > 0 && Wo > 0 && n_branches >= 1 && n_branches <= 4 && T && ps, "bad shuffle arguments");
    ShuffleBranches br;
    br.n = n_branches;
    for (int b = 0; b < n_branches; ++b) {
        LISEC_CHECK_ARG(T[b] && ps[b] >= 1 && Ho % ps[b] == 0 && Wo % ps[b] == 0, "branch %d: stride must divide the map", b);
        br.ps[b] = ps[b]; br.T[b] = T[b];
    }
    const int blocks = cdiv((long long)Ho * Wo * (kHeadCols / 4), 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (backward) LISEC_LAUNCH(k_head_shuffle<true>, dim3(blocks), dim3(256), 0, st, head, Ho, Wo, br);
    else LISEC_LAUNCH(k_head_shuffle<false>, dim3(blocks), dim3(256), 0, st, head, Ho, Wo, br);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
