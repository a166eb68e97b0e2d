// BatchNormalization statistic finalisation (tiny, latency-bound kernels).
#include "bn.h"
#include "conv.h"

namespace lisec {
namespace {

// Deterministic parallel column sums of parts[nparts][ld] (double): a 1024-thread block owns 16
// consecutive columns; 64 row groups stride over the parts (loads batched 8 deep, added in index order),
// then a fixed-order LDS tree combines them.  column_sum2 sums columns `col` and `col + off` in one pass.
constexpr int kRedCols = 16, kRedRows = 64, kRedBatch = 8;

__device__ __forceinline__ void column_sum2(const double* __restrict__ parts, int nparts, size_t ld, int col,
                                            size_t off, bool two, bool col_ok, double (*red)[kRedCols],
                                            double& r1, double& r2) {
    const int cx = threadIdx.x % kRedCols, ry = threadIdx.x / kRedCols;
    double s1 = 0.0, s2 = 0.0;
    if (col_ok) {
        const double* src = parts + col;
        int b = ry;
        for (; b + (kRedBatch - 1) * kRedRows < nparts; b += kRedBatch * kRedRows) {
            double v[kRedBatch], w[kRedBatch];
#pragma unroll
            for (int u = 0; u < kRedBatch; ++u) {
                const double* q = src + (size_t)(b + u * kRedRows) * ld;
                v[u] = q[0];
                w[u] = two ? q[off] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kRedBatch; ++u) { s1 += v[u]; s2 += w[u]; }
        }
        for (; b < nparts; b += kRedRows) {
            const double* q = src + (size_t)b * ld;
            s1 += q[0];
            if (two) s2 += q[off];
        }
    }
    red[ry][cx] = s1;
    red[kRedRows + ry][cx] = s2;
    __syncthreads();
    for (int o = kRedRows / 2; o > 0; o >>= 1) {
        if (ry < o) {
            red[ry][cx] += red[ry + o][cx];
            red[kRedRows + ry][cx] += red[kRedRows + ry + o][cx];
        }
        __syncthreads();
    }
    r1 = red[0][cx];
    r2 = red[kRedRows][cx];
    __syncthreads();
}

__global__ void __launch_bounds__(1024)
k_bn_finalize(const double* __restrict__ parts, int nparts, int C, double N,
              const float* __restrict__ gamma, const float* __restrict__ beta,
              float* __restrict__ mmean, float* __restrict__ mvar, int unbiased,
              float* __restrict__ st) {
    __shared__ double red[2 * kRedRows][kRedCols];
    const int c = blockIdx.x * kRedCols + threadIdx.x % kRedCols;
    const bool ok = c < C;
    double s1, s2;
    column_sum2(parts, nparts, (size_t)2 * C, c, (size_t)C, true, ok, red, s1, s2);
    if (!ok || threadIdx.x >= kRedCols) return;
    double mean = s1 / N;
    double var = s2 / N - mean * mean;            // biased; fp64 so the cancellation is harmless
    if (var < 0.0) var = 0.0;
    double inv = 1.0 / sqrt(var + (double)kBnEps);
    double scale = (double)gamma[c] * inv;
    st[c] = (float)scale;
    st[C + c] = (float)((double)beta[c] - mean * scale);
    st[2 * C + c] = (float)mean;
    st[3 * C + c] = (float)inv;
    if (mmean) {
        double v = unbiased && N > 1.0 ? var * (N / (N - 1.0)) : var;
        mmean[c] = (float)((double)mmean[c] * (double)kBnMomentum + mean * (1.0 - (double)kBnMomentum));
        mvar[c] = (float)((double)mvar[c] * (double)kBnMomentum + v * (1.0 - (double)kBnMomentum));
    }
}

__global__ void k_bn_fold(const float* __restrict__ gamma, const float* __restrict__ beta,
                          const float* __restrict__ mmean, const float* __restrict__ mvar, int C,
                          float* __restrict__ st) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double inv = 1.0 / sqrt((double)mvar[c] + (double)kBnEps);
    double scale = (double)gamma[c] * inv;
    st[c] = (float)scale;
    st[C + c] = (float)((double)beta[c] - (double)mmean[c] * scale);
    st[2 * C + c] = mmean[c];
    st[3 * C + c] = (float)inv;
}

__global__ void __launch_bounds__(1024)
k_reduce_parts(const double* __restrict__ parts, int nparts, int C, double scale,
               float* __restrict__ out_f, double* __restrict__ out_d) {
    __shared__ double red[2 * kRedRows][kRedCols];
    const int c = blockIdx.x * kRedCols + threadIdx.x % kRedCols;
    const bool ok = c < C;
    double s, unused;
    column_sum2(parts, nparts, (size_t)C, c, 0, false, ok, red, s, unused);
    if (!ok || threadIdx.x >= kRedCols) return;
    s *= scale;
    if (out_f) out_f[c] = (float)s;
    if (out_d) out_d[c] = s;
}

__global__ void __launch_bounds__(1024)
k_bn_bwd_finalize(const double* __restrict__ parts, int nparts, int C, double N,
                  float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double red[2 * kRedRows][kRedCols];
    const int c = blockIdx.x * kRedCols + threadIdx.x % kRedCols;
    const bool ok = c < C;
    double s1, s2;
    column_sum2(parts, nparts, (size_t)2 * C, c, (size_t)C, true, ok, red, s1, s2);
    if (!ok || threadIdx.x >= kRedCols) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / N);          // mean(dz)
    coef[C + c] = (float)(s2 / N);      // mean(dz * yhat)
}

// k_reduce_parts and k_bn_bwd_finalize of ONE producer in one launch (they read different partial tables of the same
// kernel and nothing of each other): blocks [0, nb_red) run the first, the rest the second -- the same blocks, the same
// summation order, one dependent launch less on the VFE backward's chain of small kernels.
__global__ void __launch_bounds__(1024)
k_reduce_and_bwd_finalize(const double* __restrict__ rparts, int rnparts, int rC, double scale, float* __restrict__ out_f,
                          int nb_red, const double* __restrict__ parts, int nparts, int C, double N,
                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double red[2 * kRedRows][kRedCols];
    if ((int)blockIdx.x < nb_red) {
        const int c = blockIdx.x * kRedCols + threadIdx.x % kRedCols;
        const bool ok = c < rC;
        double s, unused;
        column_sum2(rparts, rnparts, (size_t)rC, c, 0, false, ok, red, s, unused);
        if (!ok || threadIdx.x >= kRedCols) return;
        out_f[c] = (float)(s * scale);
        return;
    }
    const int c = (blockIdx.x - nb_red) * kRedCols + threadIdx.x % kRedCols;
    const bool ok = c < C;
    double s1, s2;
    column_sum2(parts, nparts, (size_t)2 * C, c, (size_t)C, true, ok, red, s1, s2);
    if (!ok || threadIdx.x >= kRedCols) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / N);
    coef[C + c] = (float)(s2 / N);
}

}  // namespace

int launch_reduce_and_bwd_finalize(const double* rparts, int rnparts, int rC, double scale, float* out_f,
                                   const double* parts, int nparts, int C, double N, float* dgamma, float* dbeta,
                                   float* coef, hipStream_t st) {
    const int nb_red = cdiv(rC, kRedCols);
    LISEC_LAUNCH(k_reduce_and_bwd_finalize, dim3(nb_red + cdiv(C, kRedCols)), dim3(1024), 0, st, rparts, rnparts, rC, scale,
                 out_f, nb_red, parts, nparts, C, N, dgamma, dbeta, coef);
    LISEC_LAUNCH_CHECK();
    return 0;
}

int launch_bn_bwd_finalize(const double* parts, int nparts, int C, double N, float* dgamma, float* dbeta,
                           float* coef, hipStream_t st) {
    LISEC_LAUNCH(k_bn_bwd_finalize, dim3(cdiv(C, kRedCols)), dim3(1024), 0, st, parts, nparts, C, N, dgamma,
                       dbeta, coef);
    LISEC_LAUNCH_CHECK();
    return 0;
}

int launch_bn_finalize(const double* partials, int nparts, int C, double N, const float* gamma,
                       const float* beta, float* moving_mean, float* moving_var, int unbiased_moving,
                       float* bnstate, hipStream_t st) {
    LISEC_LAUNCH(k_bn_finalize, dim3(cdiv(C, kRedCols)), dim3(1024), 0, st, partials, nparts, C, N, gamma,
                       beta, moving_mean, moving_var, unbiased_moving, bnstate);
    LISEC_LAUNCH_CHECK();
    return 0;
}

int launch_bn_fold(const float* gamma, const float* beta, const float* moving_mean,
                   const float* moving_var, int C, float* bnstate, hipStream_t st) {
    LISEC_LAUNCH(k_bn_fold, dim3(cdiv(C, 64)), dim3(64), 0, st, gamma, beta, moving_mean,
                       moving_var, C, bnstate);
    LISEC_LAUNCH_CHECK();
    return 0;
}

int launch_reduce_parts(const double* parts, int nparts, int C, double scale, float* out_f, double* out_d,
                        hipStream_t st) {
    LISEC_LAUNCH(k_reduce_parts, dim3(cdiv(C, kRedCols)), dim3(1024), 0, st, parts, nparts, C, scale,
                       out_f, out_d);
    LISEC_LAUNCH_CHECK();
    return 0;
}

}  // namespace lisec

using namespace lisec;

extern "C" int lisec_bn_finalize(const double* partials, int nparts, int C, double n_rows, const float* gamma,
                                 const float* beta, float* moving_mean, float* moving_var, int unbiased_moving,
                                 float* bnstate, lisec_stream_t stream) {
    LISEC_CHECK_ARG(partials && gamma && beta && bnstate && nparts > 0 && C > 0 && n_rows > 0, "bad bn_finalize arguments");
    LISEC_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "moving_mean/moving_var must both be set or NULL");
    return launch_bn_finalize(partials, nparts, C, n_rows, gamma, beta, moving_mean, moving_var, unbiased_moving,
                              bnstate, static_cast<hipStream_t>(stream));
}

extern "C" size_t lisec_bn_sink_words(int C) { return C > 0 ? bn_sink_words(C) : 0; }

extern "C" int lisec_bn_fold(const float* gamma, const float* beta, const float* moving_mean,
                             const float* moving_var, int C, float* bnstate, lisec_stream_t stream) {
    LISEC_CHECK_ARG(gamma && beta && moving_mean && moving_var && bnstate && C > 0, "bad bn_fold arguments");
    return launch_bn_fold(gamma, beta, moving_mean, moving_var, C, bnstate, static_cast<hipStream_t>(stream));
}
