// HBM-bound helper kernels of the training step (gfx950): BatchNormalization/ReLU backward,
// column sums (bias gradients), the two losses and the SGD-Nesterov update.
// All are streaming kernels: 16-byte accesses, coalesced along the channel axis, reductions as
// per-block partials summed in index order (deterministic; no float atomics).
#include "bn.h"

namespace lisec {
namespace {

constexpr int kEwBlocks = 1024;
constexpr int kEwThreads = 256;

// ---- BatchNormalization (+ReLU) backward ----------------------------------------------------------
// forward (consumer side): z = y*scale + shift ; a = relu ? max(z,0) : z
// pass 1: dz = dA * (relu ? z > 0 : 1);  parts[b][0][c] = sum dz,  parts[b][1][c] = sum dz*yhat
template <bool RELU>
__global__ void __launch_bounds__(kEwThreads)
k_bn_bwd_reduce(const float* __restrict__ dA, int da_stride, const float* __restrict__ y,
                const float* __restrict__ st, long long M, int C, double* __restrict__ parts) {
    __shared__ float red[2][kEwThreads][4];
    const int cq = C / 4;                           // float4 groups per row
    const int q = threadIdx.x % cq, rsub = threadIdx.x / cq, rows_per_iter = kEwThreads / cq;
    const float4 sc = reinterpret_cast<const float4*>(st)[q];
    const float4 sh = reinterpret_cast<const float4*>(st + C)[q];
    const float4 mu = reinterpret_cast<const float4*>(st + 2 * C)[q];
    const float4 is = reinterpret_cast<const float4*>(st + 3 * C)[q];
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    for (long long r = (long long)blockIdx.x * rows_per_iter + rsub; r < M; r += (long long)gridDim.x * rows_per_iter) {
        const float4 g = *reinterpret_cast<const float4*>(dA + r * da_stride + q * 4);
        const float4 v = *reinterpret_cast<const float4*>(y + r * C + q * 4);
#define LISEC_ACC(f)                                                         \
        {                                                                    \
            float dz = g.f;                                                  \
            if (RELU && !(fmaf(v.f, sc.f, sh.f) > 0.f)) dz = 0.f;            \
            s1.f += dz;                                                      \
            s2.f = fmaf(dz, (v.f - mu.f) * is.f, s2.f);                      \
        }
        LISEC_ACC(x) LISEC_ACC(y) LISEC_ACC(z) LISEC_ACC(w)
#undef LISEC_ACC
    }
    red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[0][threadIdx.x][2] = s1.z; red[0][threadIdx.x][3] = s1.w;
    red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y; red[1][threadIdx.x][2] = s2.z; red[1][threadIdx.x][3] = s2.w;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += kEwThreads) {
        const int which = i / C, c = i % C;
        double a = 0.0;
        for (int k = 0; k < rows_per_iter; ++k) a += (double)red[which][k * cq + c / 4][c % 4];
        parts[((size_t)blockIdx.x * 2 + which) * C + c] = a;
    }
}

// pass 2: dy = scale * (dz - mean(dz) - yhat * mean(dz*yhat));  parts[b][c] = sum dy (conv bias gradient)
template <bool RELU>
__global__ void __launch_bounds__(kEwThreads)
k_bn_bwd_apply(const float* __restrict__ dA, int da_stride, const float* __restrict__ y,
               const float* __restrict__ st, const float* __restrict__ coef, long long M, int C,
               float* __restrict__ dy, double* __restrict__ parts) {
    __shared__ float red[kEwThreads][4];
    const int cq = C / 4;
    const int q = threadIdx.x % cq, rsub = threadIdx.x / cq, rows_per_iter = kEwThreads / cq;
    const float4 sc = reinterpret_cast<const float4*>(st)[q];
    const float4 sh = reinterpret_cast<const float4*>(st + C)[q];
    const float4 mu = reinterpret_cast<const float4*>(st + 2 * C)[q];
    const float4 is = reinterpret_cast<const float4*>(st + 3 * C)[q];
    const float4 m1 = reinterpret_cast<const float4*>(coef)[q];
    const float4 m2 = reinterpret_cast<const float4*>(coef + C)[q];
    float4 s = make_float4(0, 0, 0, 0);
    for (long long r = (long long)blockIdx.x * rows_per_iter + rsub; r < M; r += (long long)gridDim.x * rows_per_iter) {
        const float4 g = *reinterpret_cast<const float4*>(dA + r * da_stride + q * 4);
        const float4 v = *reinterpret_cast<const float4*>(y + r * C + q * 4);
        float4 o;
#define LISEC_APP(f)                                                         \
        {                                                                    \
            float dz = g.f;                                                  \
            if (RELU && !(fmaf(v.f, sc.f, sh.f) > 0.f)) dz = 0.f;            \
            o.f = sc.f * (dz - m1.f - (v.f - mu.f) * is.f * m2.f);           \
            s.f += o.f;                                                      \
        }
        LISEC_APP(x) LISEC_APP(y) LISEC_APP(z) LISEC_APP(w)
#undef LISEC_APP
        *reinterpret_cast<float4*>(dy + r * C + q * 4) = o;
    }
    if (parts) {
        red[threadIdx.x][0] = s.x; red[threadIdx.x][1] = s.y; red[threadIdx.x][2] = s.z; red[threadIdx.x][3] = s.w;
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += kEwThreads) {
            double a = 0.0;
            for (int k = 0; k < rows_per_iter; ++k) a += (double)red[k * cq + c / 4][c % 4];
            parts[(size_t)blockIdx.x * C + c] = a;
        }
    }
}

__global__ void k_relu_mask(float* __restrict__ g, const float* __restrict__ u, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(g)[i];
        const float4 b = reinterpret_cast<const float4*>(u)[i];
        a.x = b.x > 0.f ? a.x : 0.f; a.y = b.y > 0.f ? a.y : 0.f;
        a.z = b.z > 0.f ? a.z : 0.f; a.w = b.w > 0.f ? a.w : 0.f;
        reinterpret_cast<float4*>(g)[i] = a;
    }
}

// column sums of x (M, C) with row stride; one thread per column per block slice
__global__ void __launch_bounds__(kEwThreads)
k_colsum(const float* __restrict__ x, int stride, long long M, int C, double* __restrict__ parts) {
    __shared__ float red[kEwThreads];
    // C <= 256 dividing 256: one slab; C a multiple of 256: blockIdx.y walks the 256-column slabs
    const int cols = C < kEwThreads ? C : kEwThreads;
    const int c0 = blockIdx.y * kEwThreads;
    const int c = threadIdx.x % cols, rsub = threadIdx.x / cols, rows_per_iter = kEwThreads / cols;
    float s = 0.f;
    if (rsub < rows_per_iter)
        for (long long r = (long long)blockIdx.x * rows_per_iter + rsub; r < M; r += (long long)gridDim.x * rows_per_iter)
            s += x[r * stride + c0 + c];
    red[threadIdx.x] = rsub < rows_per_iter ? s : 0.f;
    __syncthreads();
    if (threadIdx.x < cols) {
        double a = 0.0;
        for (int k = 0; k < rows_per_iter; ++k) a += (double)red[k * cols + threadIdx.x];
        parts[(size_t)blockIdx.x * C + c0 + threadIdx.x] = a;
    }
}

// ---- losses -----------------------------------------------------------------------------------------
// head (M,16): [:, :2] classification map, [:, 2:] regression map (model_training.py:254-255)
// kind 0: 'mse' + 'mse' (model_training.py:296): loss = mean((cls-yc)^2) + mean((reg-yr)^2)
// kind 1: sigmoid cross-entropy on the class map (targets clipped to [0,1]) + SmoothL1 (delta 1) on the
//         regression map, both as means (BASELINE config 4)
__global__ void __launch_bounds__(kEwThreads)
k_loss(const float* __restrict__ head, const float* __restrict__ ycls, const float* __restrict__ yreg,
       long long M, int kind, float gscale, float* __restrict__ dhead, double* __restrict__ parts) {
    __shared__ double red[2][kEwThreads / 64];
    double lc = 0.0, lr = 0.0;
    const float wc = gscale / (float)(M * 2), wr = gscale / (float)(M * 14);
    for (long long i = blockIdx.x * (long long)kEwThreads + threadIdx.x; i < M * 16; i += (long long)gridDim.x * kEwThreads) {
        const long long m = i >> 4;
        const int c = (int)(i & 15);
        const float p = head[i];
        float g;
        if (c < 2) {
            const float t = ycls[m * 2 + c];
            if (kind == 0) {
                const float d = p - t;
                lc += (double)d * d;
                g = 2.f * d * wc;
            } else {
                const float tt = fminf(fmaxf(t, 0.f), 1.f);
                // log(1+exp(-|p|)) + max(p,0) - p*t
                lc += (double)(fmaxf(p, 0.f) - p * tt + log1pf(expf(-fabsf(p))));
                g = (1.f / (1.f + expf(-p)) - tt) * wc;
            }
        } else {
            const float t = yreg[m * 14 + (c - 2)];
            const float d = p - t;
            if (kind == 0) {
                lr += (double)d * d;
                g = 2.f * d * wr;
            } else {
                const float ad = fabsf(d);
                lr += (double)(ad < 1.f ? 0.5f * d * d : ad - 0.5f);
                g = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * wr;
            }
        }
        dhead[i] = g;
    }
    lc = wave_sum(lc); lr = wave_sum(lr);
    if (lane_id() == 0) { red[0][threadIdx.x >> 6] = lc; red[1][threadIdx.x >> 6] = lr; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double a = 0.0;
        for (int k = 0; k < kEwThreads / 64; ++k) a += red[threadIdx.x][k];
        parts[(size_t)blockIdx.x * 2 + threadIdx.x] = a;
    }
}

__global__ void __launch_bounds__(256)
k_loss_finalize(const double* __restrict__ parts, int nparts, double M, float* __restrict__ out) {
    __shared__ double red[2][256];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) { a += parts[2 * k]; b += parts[2 * k + 1]; }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a = red[0][0] / (M * 2.0); b = red[1][0] / (M * 14.0);
        out[0] = (float)(a + b); out[1] = (float)a; out[2] = (float)b;
    }
}

// optimizers.SGD(lr, decay, momentum, nesterov=True) (model_training.py:295):
//   v <- m*v - lr_t*g ;  w <- w + m*v - lr_t*g       (lr_t = lr/(1+decay*iter), computed by the host)
__global__ void k_sgd_nesterov(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v,
                               long long n4, float lr_t, float mom) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 W = reinterpret_cast<float4*>(w)[i], V = reinterpret_cast<float4*>(v)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
#define LISEC_UPD(f) { float nv = mom * V.f - lr_t * G.f; V.f = nv; W.f = W.f + mom * nv - lr_t * G.f; }
        LISEC_UPD(x) LISEC_UPD(y) LISEC_UPD(z) LISEC_UPD(w)
#undef LISEC_UPD
        reinterpret_cast<float4*>(w)[i] = W;
        reinterpret_cast<float4*>(v)[i] = V;
    }
}

// The same update with the iteration count on the device (a captured HIP graph of the step bakes kernel arguments in):
// state[0] = iterations, state[1] = ticket (0 between launches).  Every workgroup reads state[0] before it takes its
// ticket; the workgroup that takes the last one increments the count, i.e. after all of them have read it.
// advance == 0: the update of a PART of the variables ahead of the rest (same iteration count: it is not incremented).
__global__ void k_sgd_nesterov_dev(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v,
                                   long long n4, double lr, double decay, float mom, long long* __restrict__ state, int advance) {
    const long long it = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float lr_t = (float)(lr / (1.0 + decay * (double)it));       // what the host computes in double, then rounds
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 W = reinterpret_cast<float4*>(w)[i], V = reinterpret_cast<float4*>(v)[i];
        const float4 G = reinterpret_cast<const float4*>(g)[i];
#define LISEC_UPD(f) { float nv = mom * V.f - lr_t * G.f; V.f = nv; W.f = W.f + mom * nv - lr_t * G.f; }
        LISEC_UPD(x) LISEC_UPD(y) LISEC_UPD(z) LISEC_UPD(w)
#undef LISEC_UPD
        reinterpret_cast<float4*>(w)[i] = W;
        reinterpret_cast<float4*>(v)[i] = V;
    }
    if (!advance) return;
    __syncthreads();                                                   // every wave of this workgroup has read `it`
    if (threadIdx.x == 0) {
        const unsigned long long t = atomicAdd(reinterpret_cast<unsigned long long*>(&state[1]), 1ULL);
        if (t == (unsigned long long)gridDim.x - 1) {
            __hip_atomic_store(&state[1], 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&state[0], it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Permute((2,3,4,1)) + Reshape of the reference (model_training.py:242-243): (D,H,W,C) -> (H,W,C*D), channel c*D + d.
// inverse: the gradient's way back, optionally gated by the ReLU of the tensor that was folded (mask > 0).
__global__ void k_fold_depth(const float* __restrict__ in, float* __restrict__ out, int D, long long HW, int C,
                             int inverse, const float* __restrict__ mask) {
    const long long total = (long long)D * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // i walks the (D,H,W,C) tensor: coalesced on that side
        const int c = (int)(i % C);
        const long long t = i / C;
        const long long p = t % HW;
        const int d = (int)(t / HW);
        const long long j = (p * C + c) * D + d;
        if (!inverse) out[j] = in[i];
        else out[i] = (!mask || mask[i] > 0.f) ? in[j] : 0.f;
    }
}

__global__ void k_scale(float* __restrict__ x, long long n4, float s) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(x)[i];
        a.x *= s; a.y *= s; a.z *= s; a.w *= s;
        reinterpret_cast<float4*>(x)[i] = a;
    }
}

int ew_blocks(long long work_items) {
    long long b = (work_items + kEwThreads - 1) / kEwThreads;
    if (b < 1) b = 1;
    return (int)(b > kEwBlocks ? kEwBlocks : b);
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" size_t lisec_eltwise_workspace_bytes(void) {
    // the largest user: bn backward pass 1, double[kEwBlocks][2][256] + coefficients
    return align_up(sizeof(double) * (size_t)kEwBlocks * 2 * 256, 256) + 4096;
}

extern "C" int lisec_bn_backward(const float* dA, int da_stride, const float* y, const float* bnstate,
                                 long long M, int C, int relu, float* dgamma, float* dbeta, float* dbias,
                                 float* dy, void* workspace, size_t workspace_bytes, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(dA && y && bnstate && dgamma && dbeta && dy && workspace, "NULL pointer");
    LISEC_CHECK_ARG(M > 0 && C >= 4 && C <= 256 && C % 4 == 0 && (kEwThreads * 4) % C == 0 && da_stride % 4 == 0,
                    "bn_backward: C must divide 1024 and be a multiple of 4");
    if (workspace_bytes < lisec_eltwise_workspace_bytes()) {
        set_error("eltwise workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    double* parts = static_cast<double*>(workspace);
    float* coef = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up(sizeof(double) * (size_t)kEwBlocks * 2 * 256, 256));
    const int rows_per_iter = kEwThreads / (C / 4);
    int nb = (int)((M + rows_per_iter - 1) / rows_per_iter);
    if (nb > kEwBlocks) nb = kEwBlocks;
    if (relu)
        LISEC_LAUNCH(k_bn_bwd_reduce<true>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, M, C, parts);
    else
        LISEC_LAUNCH(k_bn_bwd_reduce<false>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, M, C, parts);
    if (int rc = launch_bn_bwd_finalize(parts, nb, C, (double)M, dgamma, dbeta, coef, st)) return rc;
    double* bparts = dbias ? parts : nullptr;
    if (relu)
        LISEC_LAUNCH(k_bn_bwd_apply<true>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, coef, M, C, dy, bparts);
    else
        LISEC_LAUNCH(k_bn_bwd_apply<false>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, coef, M, C, dy, bparts);
    LISEC_LAUNCH_CHECK();
    if (dbias) return launch_reduce_parts(parts, nb, C, 1.0, dbias, nullptr, st);
    return LISEC_OK;
}

extern "C" int lisec_bn_backward_apply(const float* dA, int da_stride, const float* y, const float* bnstate,
                                       long long M, int C, int relu, const double* partials, int nparts,
                                       float* dgamma, float* dbeta, float* dy, void* workspace,
                                       size_t workspace_bytes, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(dA && y && bnstate && partials && dgamma && dbeta && dy && workspace && nparts > 0, "NULL pointer");
    LISEC_CHECK_ARG(M > 0 && C >= 4 && C <= 256 && C % 4 == 0 && (kEwThreads * 4) % C == 0 && da_stride % 4 == 0,
                    "bn_backward: C must divide 1024 and be a multiple of 4");
    if (workspace_bytes < lisec_eltwise_workspace_bytes()) {
        set_error("eltwise workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    float* coef = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up(sizeof(double) * (size_t)kEwBlocks * 2 * 256, 256));
    if (int rc = launch_bn_bwd_finalize(partials, nparts, C, (double)M, dgamma, dbeta, coef, st)) return rc;
    const int rows_per_iter = kEwThreads / (C / 4);
    int nb = (int)((M + rows_per_iter - 1) / rows_per_iter);
    if (nb > kEwBlocks) nb = kEwBlocks;
    if (relu)
        LISEC_LAUNCH(k_bn_bwd_apply<true>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, coef, M, C, dy,
                           (double*)nullptr);
    else
        LISEC_LAUNCH(k_bn_bwd_apply<false>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, coef, M, C, dy,
                           (double*)nullptr);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_bn_backward_apply_coef(const float* dA, int da_stride, const float* y, const float* bnstate,
                                            long long M, int C, int relu, const float* coef, float* dy,
                                            lisec_stream_t stream_) {
    LISEC_CHECK_ARG(dA && y && bnstate && coef && dy, "NULL pointer");
    LISEC_CHECK_ARG(M > 0 && C >= 4 && C <= 256 && C % 4 == 0 && (kEwThreads * 4) % C == 0 && da_stride % 4 == 0,
                    "bn_backward: C must divide 1024 and be a multiple of 4");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const int rows_per_iter = kEwThreads / (C / 4);
    int nb = (int)((M + rows_per_iter - 1) / rows_per_iter);
    if (nb > kEwBlocks) nb = kEwBlocks;
    if (relu)
        LISEC_LAUNCH(k_bn_bwd_apply<true>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, coef, M, C, dy,
                           (double*)nullptr);
    else
        LISEC_LAUNCH(k_bn_bwd_apply<false>, dim3(nb), dim3(kEwThreads), 0, st, dA, da_stride, y, bnstate, coef, M, C, dy,
                           (double*)nullptr);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

// n small strided 2-D copies in one launch (a workgroup per descriptor): the head kernels / biases and their gradients
// move between the Keras-shaped variables and the merged (768,16) head layout without one tiny launch per slice
__global__ void k_copy2d_batched(const lisec_copy_desc* __restrict__ tab) {
    const lisec_copy_desc d = tab[blockIdx.x];
    const long long total = (long long)d.rows * d.cols;     // (public entry point: the product may exceed 2^31)
    for (long long i = blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.y * blockDim.x) {
        const long long r = i / d.cols;
        const int c = (int)(i - r * d.cols);
        d.dst[r * d.dst_stride + c] = d.src[r * d.src_stride + c];
    }
}

extern "C" int lisec_copy2d_batched(const lisec_copy_desc* device_table, int n, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(device_table && n > 0, "bad batched copy arguments");
    LISEC_LAUNCH(k_copy2d_batched, dim3(n, 16), dim3(256), 0, static_cast<hipStream_t>(stream_), device_table);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_relu_mask(float* grad, const float* act, long long n, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(grad && act && n >= 0 && n % 4 == 0, "relu_mask: n must be a multiple of 4");
    if (n == 0) return LISEC_OK;
    LISEC_LAUNCH(k_relu_mask, dim3(ew_blocks(n / 4)), dim3(kEwThreads), 0, static_cast<hipStream_t>(stream_),
                       grad, act, n / 4);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_colsum(const float* x, int stride, long long M, int C, float* out, void* workspace,
                            size_t workspace_bytes, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(x && out && workspace && M > 0 && C >= 1 && stride >= C &&
                    ((C <= 256 && kEwThreads % C == 0) || (C % 256 == 0 && C <= 1024)),
                    "colsum: C must divide 256 or be a multiple of 256 up to 1024");
    if (workspace_bytes < lisec_eltwise_workspace_bytes()) {
        set_error("eltwise workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    double* parts = static_cast<double*>(workspace);
    const int rows_per_iter = C < kEwThreads ? kEwThreads / C : 1;
    const int slabs = C < kEwThreads ? 1 : C / kEwThreads;
    int nb = (int)((M + rows_per_iter - 1) / rows_per_iter);
    const int cap = kEwBlocks * 2 * 256 / C;                 // partial rows the workspace holds
    if (nb > cap) nb = cap;
    if (nb > kEwBlocks) nb = kEwBlocks;
    LISEC_LAUNCH(k_colsum, dim3(nb, slabs), dim3(kEwThreads), 0, st, x, stride, M, C, parts);
    LISEC_LAUNCH_CHECK();
    return launch_reduce_parts(parts, nb, C, 1.0, out, nullptr, st);
}

extern "C" int lisec_rpn_loss(const float* head, const float* y_cls, const float* y_reg, long long M, int kind,
                              float grad_scale, float* dhead, float* loss_out, void* workspace,
                              size_t workspace_bytes, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(head && y_cls && y_reg && dhead && loss_out && workspace && M > 0, "NULL pointer");
    LISEC_CHECK_ARG(kind == 0 || kind == 1, "loss kind must be 0 (mse+mse) or 1 (sigmoid-CE + SmoothL1)");
    if (workspace_bytes < lisec_eltwise_workspace_bytes()) {
        set_error("eltwise workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    double* parts = static_cast<double*>(workspace);
    int nb = ew_blocks(M * 16);
    LISEC_LAUNCH(k_loss, dim3(nb), dim3(kEwThreads), 0, st, head, y_cls, y_reg, M, kind, grad_scale, dhead, parts);
    LISEC_LAUNCH(k_loss_finalize, dim3(1), dim3(256), 0, st, parts, nb, (double)M, loss_out);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_sgd_nesterov_step(float* theta, const float* grad, float* velocity, long long n,
                                       float lr_t, float momentum, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(theta && grad && velocity && n >= 0 && n % 4 == 0, "sgd: n must be a multiple of 4");
    if (n == 0) return LISEC_OK;
    LISEC_LAUNCH(k_sgd_nesterov, dim3(ew_blocks(n / 4)), dim3(kEwThreads), 0, static_cast<hipStream_t>(stream_),
                       theta, grad, velocity, n / 4, lr_t, momentum);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_sgd_nesterov_step_dev(float* theta, const float* grad, float* velocity, long long n, double lr,
                                           double decay, float momentum, long long* state, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(theta && grad && velocity && state && n >= 0 && n % 4 == 0, "sgd: n must be a multiple of 4");
    if (n == 0) return LISEC_OK;
    LISEC_LAUNCH(k_sgd_nesterov_dev, dim3(ew_blocks(n / 4)), dim3(kEwThreads), 0, static_cast<hipStream_t>(stream_),
                       theta, grad, velocity, n / 4, lr, decay, momentum, state, 1);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_sgd_nesterov_step_dev_part(float* theta, const float* grad, float* velocity, long long n, double lr,
                                                double decay, float momentum, const long long* state, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(theta && grad && velocity && state && n >= 0 && n % 4 == 0, "sgd: n must be a multiple of 4");
    if (n == 0) return LISEC_OK;
    LISEC_LAUNCH(k_sgd_nesterov_dev, dim3(ew_blocks(n / 4)), dim3(kEwThreads), 0, static_cast<hipStream_t>(stream_),
                       theta, grad, velocity, n / 4, lr, decay, momentum, const_cast<long long*>(state), 0);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_fold_depth(const float* in, float* out, int D, long long HW, int C, int inverse, const float* mask,
                                lisec_stream_t stream_) {
    LISEC_CHECK_ARG(in && out && D >= 1 && HW >= 1 && C >= 1, "bad fold arguments");
    LISEC_LAUNCH(k_fold_depth, dim3(ew_blocks((long long)D * HW * C)), dim3(kEwThreads), 0,
                       static_cast<hipStream_t>(stream_), in, out, D, HW, C, inverse, mask);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_scale(float* x, long long n, float s, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(x && n >= 0 && n % 4 == 0, "scale: n must be a multiple of 4");
    if (n == 0) return LISEC_OK;
    LISEC_LAUNCH(k_scale, dim3(ew_blocks(n / 4)), dim3(kEwThreads), 0, static_cast<hipStream_t>(stream_), x, n / 4, s);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
