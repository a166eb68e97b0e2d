// Internal: BatchNormalization statistics helpers shared by the VFE and conv kernels.
// Keras semantics (model_training.py:171,194,204): axis -1, eps 1e-3, momentum 0.99; training
// normalises with the batch mean and BIASED batch variance over every other axis.
#pragma once
#include "common.h"

namespace lisec {

constexpr float kBnEps = 1e-3f;
constexpr float kBnMomentum = 0.99f;

// "bnstate": float[4][C] = scale, shift, mean, invstd with
//   scale = gamma*rsqrt(var+eps), shift = beta - mean*scale   (y*scale + shift == BN(y))
struct BnState {
    float* p;
    int C;
    __host__ __device__ float* scale() const { return p; }
    __host__ __device__ float* shift() const { return p + C; }
    __host__ __device__ float* mean() const { return p + 2 * C; }
    __host__ __device__ float* invstd() const { return p + 3 * C; }
};

// partials: double[nparts][2][C] = (sum w*y, sum w*y*y); N = number of dense rows reduced over.
// Sums the parts in index order (deterministic), writes bnstate, and (moving_* != NULL) updates
// the moving statistics in place: m <- m*0.99 + batch*0.01, variance Bessel-corrected when
// `unbiased_moving` (Keras fused rank-4/5 path) and biased otherwise (rank-6 VFE path).
int launch_bn_finalize(const double* partials, int nparts, int C, double N, const float* gamma,
                       const float* beta, float* moving_mean, float* moving_var, int unbiased_moving,
                       float* bnstate, hipStream_t st);

// inference: bnstate from the moving statistics
int launch_bn_fold(const float* gamma, const float* beta, const float* moving_mean,
                   const float* moving_var, int C, float* bnstate, hipStream_t st);

// out[c] = sum_b parts[b][c] (index order); out_f and/or out_d may be NULL; `scale` multiplies the sum
int launch_reduce_parts(const double* parts, int nparts, int C, double scale, float* out_f, double* out_d,
                        hipStream_t st);

// launch_reduce_parts(rparts -> out_f) and launch_bn_bwd_finalize(parts -> dgamma, dbeta, coef) in ONE launch
int launch_reduce_and_bwd_finalize(const double* rparts, int rnparts, int rC, double scale, float* out_f,
                                   const double* parts, int nparts, int C, double N, float* dgamma, float* dbeta,
                                   float* coef, hipStream_t st);

// BN backward finalisation: parts double[nparts][2][C] = (sum dz, sum dz*yhat) ->
// dbeta, dgamma (fp32 gradients) and coef float[2][C] = (mean dz, mean dz*yhat) over N dense rows
int launch_bn_bwd_finalize(const double* parts, int nparts, int C, double N, float* dgamma, float* dbeta,
                           float* coef, hipStream_t st);

}  // namespace lisec
