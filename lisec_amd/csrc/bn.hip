// BatchNormalization statistic finalisation (tiny, latency-bound kernels).
#include "bn.h"

namespace lisec {
namespace {

// Deterministic parallel column sums of parts[nparts][ld] (double): a 1024-thread block owns 32
// consecutive columns; 32 row groups stride over the parts, then a fixed-order LDS tree combines them.
constexpr int kRedCols = 32, kRedRows = 32;

__device__ __forceinline__ double column_sum(const double* __restrict__ parts, int nparts, size_t ld, int col,
                                             bool col_ok, double (*red)[kRedCols]) {
    const int cx = threadIdx.x % kRedCols, ry = threadIdx.x / kRedCols;
    double s = 0.0;
    if (col_ok)
        for (int b = ry; b < nparts; b += kRedRows) s += parts[(size_t)b * ld + col];
    red[ry][cx] = s;
    __syncthreads();
    for (int o = kRedRows / 2; o > 0; o >>= 1) {
        if (ry < o) red[ry][cx] += red[ry + o][cx];
        __syncthreads();
    }
    const double r = red[0][cx];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(1024)
k_bn_finalize(const double* __restrict__ parts, int nparts, int C, double N,
              const float* __restrict__ gamma, const float* __restrict__ beta,
              float* __restrict__ mmean, float* __restrict__ mvar, int unbiased,
              float* __restrict__ st) {
    __shared__ double red[kRedRows][kRedCols];
    const int c = blockIdx.x * kRedCols + threadIdx.x % kRedCols;
    const bool ok = c < C;
    const double s1 = column_sum(parts, nparts, (size_t)2 * C, c, ok, red);
    const double s2 = column_sum(parts + C, nparts, (size_t)2 * C, c, ok, red);
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
    __shared__ double red[kRedRows][kRedCols];
    const int c = blockIdx.x * kRedCols + threadIdx.x % kRedCols;
    const bool ok = c < C;
    double s = column_sum(parts, nparts, (size_t)C, c, ok, red);
    if (!ok || threadIdx.x >= kRedCols) return;
    s *= scale;
    if (out_f) out_f[c] = (float)s;
    if (out_d) out_d[c] = s;
}

__global__ void __launch_bounds__(1024)
k_bn_bwd_finalize(const double* __restrict__ parts, int nparts, int C, double N,
                  float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double red[kRedRows][kRedCols];
    const int c = blockIdx.x * kRedCols + threadIdx.x % kRedCols;
    const bool ok = c < C;
    const double s1 = column_sum(parts, nparts, (size_t)2 * C, c, ok, red);
    const double s2 = column_sum(parts + C, nparts, (size_t)2 * C, c, ok, red);
    if (!ok || threadIdx.x >= kRedCols) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / N);          // mean(dz)
    coef[C + c] = (float)(s2 / N);      // mean(dz * yhat)
}

}  // namespace

int launch_bn_bwd_finalize(const double* parts, int nparts, int C, double N, float* dgamma, float* dbeta,
                           float* coef, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(cdiv(C, kRedCols)), dim3(1024), 0, st, parts, nparts, C, N, dgamma,
                       dbeta, coef);
    LISEC_LAUNCH_CHECK();
    return 0;
}

int launch_bn_finalize(const double* partials, int nparts, int C, double N, const float* gamma,
                       const float* beta, float* moving_mean, float* moving_var, int unbiased_moving,
                       float* bnstate, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, kRedCols)), dim3(1024), 0, st, partials, nparts, C, N, gamma,
                       beta, moving_mean, moving_var, unbiased_moving, bnstate);
    LISEC_LAUNCH_CHECK();
    return 0;
}

int launch_bn_fold(const float* gamma, const float* beta, const float* moving_mean,
                   const float* moving_var, int C, float* bnstate, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_fold, dim3(cdiv(C, 64)), dim3(64), 0, st, gamma, beta, moving_mean,
                       moving_var, C, bnstate);
    LISEC_LAUNCH_CHECK();
    return 0;
}

int launch_reduce_parts(const double* parts, int nparts, int C, double scale, float* out_f, double* out_d,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_parts, dim3(cdiv(C, kRedCols)), dim3(1024), 0, st, parts, nparts, C, scale,
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

extern "C" int lisec_bn_fold(const float* gamma, const float* beta, const float* moving_mean,
                             const float* moving_var, int C, float* bnstate, lisec_stream_t stream) {
    LISEC_CHECK_ARG(gamma && beta && moving_mean && moving_var && bnstate && C > 0, "bad bn_fold arguments");
    return launch_bn_fold(gamma, beta, moving_mean, moving_var, C, bnstate, static_cast<hipStream_t>(stream));
}
