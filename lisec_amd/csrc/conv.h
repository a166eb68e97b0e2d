// Internal: geometry shared by the implicit-GEMM kernels (igemm.hip, wgrad.hip).
#pragma once
#include "common.h"

namespace lisec {

// Where a contraction's per-tile BatchNormalization sums go when the caller passes a lisec_bn_sink: order-independent
// fixed-point accumulators (common.h) instead of a partial table, and the LAST workgroup of the call to arrive
// finalises them -- no finaliser launch.  acc: long long[kSinkReplicas][2][C][2 limbs] + a ticket word.
constexpr int kSinkReplicas = 4;
struct BnSink {
    long long* acc;          // nullptr: off
    int kind, C, unbiased;   // kind 1: forward batch statistics -> bnstate (+ moving statistics); 2: backward -> coef
    unsigned total;          // workgroups that feed the sink in this call (all launches together)
    double N;                // dense rows the statistics are over
    const float* gamma; const float* beta; float* mmean; float* mvar; float* bnstate;
    float* dgamma; float* dbeta; float* coef;
};
inline size_t bn_sink_words(int C) { return (size_t)kSinkReplicas * 2 * C * 2 + 2; }

struct ConvGeom {
    int Di, Hi, Wi;          // tensor that is gathered from
    int Do, Ho, Wo;          // tensor that is written (M = Do*Ho*Wo rows)
    int KD, KH, KW;
    int ls_d, ls_h, ls_w;    // log2(stride)
    int pd, ph, pw;
    int Cin, in_stride, Cout, out_stride, CoutP;
    int M;
    int ps, ps_channels;
    // optional row list: GEMM row m is the position (d,h,w) = row_coords[3m..3m+2]; rows >= *row_count are void
    const int* row_coords;
    const int* row_count;
    // optional output mask (same layout as out): stored value = mask > 0 ? value : 0 (ReLU gradient gate)
    const float* out_mask;
    // parity-class row order (igemm, mode 1 with stride 2 along h and w, 2D): GEMM row m = class * pc_span + q,
    // class = (h & 1) * 2 + (w & 1), q = (h >> 1) * (Wo / 2) + (w >> 1) < pc_rows; pc_span = rows per class rounded
    // up to whole tiles, M = 4 * pc_span.  Every tile then has ONE parity, i.e. one set of taps that divide.
    int pc_span, pc_rows;
    int pointwise;           // 1x1x1 taps, unit strides, no padding, dense rows: row m reads position m
    // plane_pair: a workgroup of k_igemm_halo runs the tiles of one (h, w) place of depth planes 2q and 2q + 1 (planes are
    // whole numbers of tiles, plane_tiles each; the grid has half as many workgroups) -- equal work per workgroup when the
    // planes of a strided / transposed Conv3D run different numbers of taps
    int plane_tiles, plane_pair;
    // row lists over a big capacity: k_igemm's resident workgroups draw tiles from queue[0] (see k_igemm); else nullptr
    int* queue;
    // optional BatchNormalization-backward statistics of the stored gradient (see lisec_conv_extras): the per-tile
    // partials become (sum dz, sum dz*yhat) with dz = stored value * (relu ? bn(y) > 0 : 1), yhat = (y - mean)*invstd
    const float* bwd_y;      // (positions, Cout) raw conv output of the layer the gradient belongs to, row stride Cout
    const float* bwd_bn;     // its bnstate float[4*Cout]
    int bwd_relu;
    BnSink sink;             // optional destination of the per-tile sums (acc == nullptr: the partial table, if any)
    // optional second contraction on the stored tile (lisec_conv_extras.tail_w): tail_out[m, :] = stored out[m, :] @ tail_w,
    // 64 -> 64; the backward statistics / sink then belong to tail_out
    const float* tail_w;
    float* tail_out;
    // optional BatchNormalization-backward apply ON LOAD of the gathered operand (lisec_conv_extras.in_y): see igemm_tile FOLD
    const float* in_y;       // raw output of the layer the gathered gradient belongs to, laid out like `in`
    const float* fold_bn;    // its bnstate float[4*Cin]
    const float* fold_coef;  // float[2*Cin]: mean(dz), mean(dz * yhat) (a backward sink's coef)
    int fold_relu;
    // optional weight gradient of the Dense(64) whose data gradient this call is (lisec_conv_extras.dense_dw): per-workgroup
    // 64 x 64 slabs of  sum_m bn(bwd_y)[m, i] * in[m, j]  (k_dense64<false, true, true>)
    float* dw_slabs;
};

int conv_geom_check(const lisec_conv_geom* c, ConvGeom* g);

#ifdef __HIPCC__
__device__ __forceinline__ void sink_add(const BnSink& s, int which, int ch, double v) {
    if (v != 0.0) fx_atomic_add(s.acc + ((size_t)((blockIdx.x % kSinkReplicas) * 2 + which) * s.C + ch) * 2, v);
}

// Called by EVERY thread of the workgroup once its sink_add calls are issued.  The adds are waited for (vmcnt covers
// atomics), the workgroup takes a ticket, and the one that takes the last ticket of the call reads the totals
// (everything went through device-scope atomics: no fence needed), writes the BatchNormalization state or the
// backward coefficients -- the arithmetic of k_bn_finalize / k_bn_bwd_finalize -- and leaves the accumulators zeroed.
__device__ __forceinline__ void sink_finish(const BnSink& s) {
    __shared__ int sink_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(s.acc + (size_t)kSinkReplicas * 2 * s.C * 2);
    if (threadIdx.x == 0) sink_last = atomicAdd(ticket, 1ULL) == (unsigned long long)s.total - 1;
    __syncthreads();
    if (!sink_last) return;
    for (int c = threadIdx.x; c < s.C; c += blockDim.x) {
        // every limb of every replica is requested before the first one is used: one round trip for the lot (read one
        // after the other the 4 x 2 x 2 loads cost ~1 us per replica at the tail of every call that feeds a sink)
        long long hi[2][kSinkReplicas], lo[2][kSinkReplicas];
        float gam = 0.f, bet = 0.f, mm = 0.f, mv = 0.f;    // (and the per-channel parameters the result needs)
        if (s.kind == 1) {
            gam = s.gamma[c]; bet = s.beta[c];
            if (s.mmean) { mm = s.mmean[c]; mv = s.mvar[c]; }
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int r = 0; r < kSinkReplicas; ++r) {
                long long* p = s.acc + ((size_t)(r * 2 + which) * s.C + c) * 2;
                hi[which][r] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo[which][r] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        double v[2];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            long long h = 0, l = 0;
#pragma unroll
            for (int r = 0; r < kSinkReplicas; ++r) {
                h += hi[which][r];
                l += lo[which][r];
                long long* p = s.acc + ((size_t)(r * 2 + which) * s.C + c) * 2;
                __hip_atomic_store(p, 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p + 1, 0LL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            v[which] = fx_join(h, l);
        }
        if (s.kind == 1) {
            const double mean = v[0] / s.N;
            double var = v[1] / s.N - mean * mean;        // biased; fp64 so the cancellation is harmless
            if (var < 0.0) var = 0.0;
            const double inv = 1.0 / sqrt(var + 1e-3);    // kBnEps
            const double scale = (double)gam * inv;
            s.bnstate[c] = (float)scale;
            s.bnstate[s.C + c] = (float)((double)bet - mean * scale);
            s.bnstate[2 * s.C + c] = (float)mean;
            s.bnstate[3 * s.C + c] = (float)inv;
            if (s.mmean) {
                const double u = s.unbiased && s.N > 1.0 ? var * (s.N / (s.N - 1.0)) : var;
                s.mmean[c] = (float)((double)mm * 0.99 + mean * (1.0 - 0.99));       // kBnMomentum
                s.mvar[c] = (float)((double)mv * 0.99 + u * (1.0 - 0.99));
            }
        } else {
            s.dbeta[c] = (float)v[0];
            s.dgamma[c] = (float)v[1];
            s.coef[c] = (float)(v[0] / s.N);              // mean(dz)
            s.coef[s.C + c] = (float)(v[1] / s.N);        // mean(dz * yhat)
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

#ifdef __HIPCC__
// rows that exist: M, or the device-side count of a row list
__device__ __forceinline__ int row_limit(const ConvGeom& g) {
    if (!g.row_coords) return g.M;
    const int n = *g.row_count;
    return n < g.M ? n : g.M;
}
#endif

#ifdef __HIPCC__
// parity-class row -> (h, w); false for the padding rows at the end of a class
__device__ __forceinline__ bool parity_decode(const ConvGeom& g, int m, int& h, int& w) {
    const int c = m / g.pc_span, q = m - c * g.pc_span;
    const int Wh = g.Wo >> 1;
    const int i = q / Wh, j = q - i * Wh;
    h = 2 * i + (c >> 1);
    w = 2 * j + (c & 1);
    return q < g.pc_rows;
}
#endif

#ifdef __HIPCC__
// source coordinate of output coordinate `o` for kernel tap `k` along one axis
//   mode 0: o*stride - pad + k;   mode 1: (o + pad - k)/stride when divisible
__device__ __forceinline__ int src_coord(int o, int k, int ls, int pad, int n_in, int mode, bool& ok) {
    if (mode == 0) {
        int s = (o << ls) - pad + k;
        ok = ok && s >= 0 && s < n_in;
        return s;
    }
    int t = o + pad - k;
    int s = t >> ls;
    ok = ok && t >= 0 && (t & ((1 << ls) - 1)) == 0 && s < n_in;
    return s;
}

// Strength-reduced gather addressing shared by igemm.hip / wgrad.hip.  For an output position
// (d,h,w) the source element offset of tap (kd,kh,kw) is  row_base + tap_delta  with
//   mode 0:  a = o*stride - pad           src = a + k              -> base a,          delta +k
//   mode 1:  a = o + pad                  src = a/stride - k/stride -> base a/stride,  delta -(k/stride)
//            (valid only when a - k >= 0 and divisible by the stride, for which the identity is exact)
// and the tap is valid iff the per-axis bits (kd | kh << 4 | kw << 8) are all set in the row's mask.
struct RowGather {
    int off;       // element offset of the row's base position (times in_stride), may be negative
    int mask;      // valid kd bits 0-3, kh bits 4-7, kw bits 8-11; 0 for rows beyond M
};

__device__ __forceinline__ int axis_mask(int o, int K, int ls, int pad, int n_in, int mode, int& base) {
    int m = 0;
    if (mode == 0) {
        base = (o << ls) - pad;
#pragma unroll
        for (int k = 0; k < 4; ++k) m |= (k < K && base + k >= 0 && base + k < n_in) ? (1 << k) : 0;
    } else {
        const int a = o + pad;
        base = a >> ls;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = a - k;
            m |= (k < K && t >= 0 && (t & ((1 << ls) - 1)) == 0 && (t >> ls) < n_in) ? (1 << k) : 0;
        }
    }
    return m;
}

__device__ __forceinline__ RowGather row_gather(const ConvGeom& g, int m, int mode, int lane_elem_off) {
    RowGather r;
    if (m >= g.M) { r.off = 0; r.mask = 0; return r; }
    if (g.pointwise) {                        // 1x1x1, unit stride, no padding: source position == output position
        r.off = m * g.in_stride + lane_elem_off;
        r.mask = 1 | (1 << 4) | (1 << 8);
        return r;
    }
    int d, h, w;
    if (g.row_coords) {
        if (m >= *g.row_count) { r.off = 0; r.mask = 0; return r; }
        d = g.row_coords[3 * m]; h = g.row_coords[3 * m + 1]; w = g.row_coords[3 * m + 2];
    } else if (g.pc_span) {
        d = 0;
        if (!parity_decode(g, m, h, w)) { r.off = 0; r.mask = 0; return r; }
    } else {
        const int HW = g.Ho * g.Wo;
        d = m / HW;
        const int rem = m - d * HW;
        h = rem / g.Wo; w = rem - h * g.Wo;
    }
    int bd, bh, bw;
    const int md = axis_mask(d, g.KD, g.ls_d, g.pd, g.Di, mode, bd);
    const int mh = axis_mask(h, g.KH, g.ls_h, g.ph, g.Hi, mode, bh);
    const int mw = axis_mask(w, g.KW, g.ls_w, g.pw, g.Wi, mode, bw);
    r.mask = md | (mh << 4) | (mw << 8);
    r.off = ((bd * g.Hi + bh) * g.Wi + bw) * g.in_stride + lane_elem_off;
    return r;
}

// wave-uniform per-tap quantities
__device__ __forceinline__ int tap_delta(const ConvGeom& g, int kd, int kh, int kw, int mode) {
    if (mode == 0) return ((kd * g.Hi + kh) * g.Wi + kw) * g.in_stride;
    return -((((kd >> g.ls_d) * g.Hi + (kh >> g.ls_h)) * g.Wi + (kw >> g.ls_w)) * g.in_stride);
}
__device__ __forceinline__ int tap_bits(int kd, int kh, int kw) { return (1 << kd) | (16 << kh) | (256 << kw); }

// XCD-aware bijective tile remap: tiles that are neighbours in memory stay on one XCD's L2
__device__ __forceinline__ int xcd_remap(int orig, int n) {
    const int q = n >> 3, r = n & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
#endif

}  // namespace lisec
