// Exact shortcuts for the FIRST middle layer's backward pass (gfx950).
//
// The grid the first Conv3D reads (model_training.py:235-236) is a constant vector c on every empty cell
// plus per-voxel values on the V occupied cells (vfe.hip).  Everything downstream of the grid gradient is
// linear in it and only needs (i) the gradient at the occupied cells and (ii) its SUM over the empty cells,
// and the weight gradient of that Conv3D splits the same way:
//     sum_p dGrid[p][c]      = sum_tap sum_n W[tap][c][n] * S[tap][n]
//     dW[tap][c][n]          = c[c] * S[tap][n]  +  sum_v (grid[p_v][c] - c[c]) * dy[q(p_v,tap)][n]
// with  S[tap][n] = sum of dy[m][n] over the output positions m whose tap `tap` reads inside the grid.
// So the 70.8 GFLOP dense data gradient and the 70.8 GFLOP dense weight gradient of that layer become two
// V-row contractions (row lists in igemm.hip / wgrad.hip) plus the two small kernels below -- the same
// numbers up to fp32 summation order (what Keras' autograd computes densely, model_training.py:299).
#include "conv.h"

namespace lisec {
namespace {

// per output line (d', h'): sums over w' of dy grouped by which kw taps are valid  -> line_s[line][kw][C]
// APPLY: `dy` is still the gradient in front of the layer's BatchNormalization; the apply pass of its backward
// (dy = scale * (dz - mean(dz) - yhat * mean(dz * yhat)), the arithmetic of k_bn_bwd_apply) runs on the way, the result is
// stored to dy_out (may alias dy) and summed: one pass over the 82 MB map instead of two.
template <bool APPLY>
__global__ void __launch_bounds__(256)
k_line_sums(ConvGeom g, const float* dy, float* __restrict__ line_s, const float* __restrict__ y,
            const float* __restrict__ st, const float* __restrict__ coef, float* dy_out) {
    __shared__ float red[4][256][4];
    const int C = g.Cout, cq = C / 4;
    const int q = threadIdx.x % cq, wsub = threadIdx.x / cq, wlanes = 256 / cq;
    const int line = blockIdx.x;
    float4 sc, mu, is, m1, m2;
    if (APPLY) {
        sc = reinterpret_cast<const float4*>(st)[q];
        mu = reinterpret_cast<const float4*>(st + 2 * C)[q];
        is = reinterpret_cast<const float4*>(st + 3 * C)[q];
        m1 = reinterpret_cast<const float4*>(coef)[q];
        m2 = reinterpret_cast<const float4*>(coef + C)[q];
    }
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0, 0, 0, 0);
    const size_t row0 = (size_t)line * g.Wo;
    const float* base = dy + row0 * g.out_stride;
    // kLineBatch positions per pass, every load of the pass issued before the first store: dy_out may alias dy (same
    // index only), so the compiler cannot move a later position's loads above an earlier one's store by itself, and one
    // memory round trip per position (25 for a 400-wide line) is what this kernel cost beside the weight gradients
    constexpr int kLineBatch = 5;
    for (int w0 = wsub; w0 < g.Wo; w0 += kLineBatch * wlanes) {
        float4 v[kLineBatch], yv[kLineBatch];
#pragma unroll
        for (int u = 0; u < kLineBatch; ++u) {
            const int w = w0 + u * wlanes;
            const int wc = w < g.Wo ? w : w0;                        // (clamped, not predicated)
            v[u] = *reinterpret_cast<const float4*>(base + (size_t)wc * g.out_stride + q * 4);
            if (APPLY) yv[u] = *reinterpret_cast<const float4*>(y + (row0 + wc) * C + q * 4);
        }
#pragma unroll
        for (int u = 0; u < kLineBatch; ++u) {
            const int w = w0 + u * wlanes;
            if (w < g.Wo) {
                float4 t = v[u];
                if (APPLY) {
                    t.x = sc.x * (t.x - m1.x - (yv[u].x - mu.x) * is.x * m2.x);
                    t.y = sc.y * (t.y - m1.y - (yv[u].y - mu.y) * is.y * m2.y);
                    t.z = sc.z * (t.z - m1.z - (yv[u].z - mu.z) * is.z * m2.z);
                    t.w = sc.w * (t.w - m1.w - (yv[u].w - mu.w) * is.w * m2.w);
                    *reinterpret_cast<float4*>(dy_out + (row0 + w) * g.out_stride + q * 4) = t;
                }
                const int b = (w << g.ls_w) - g.pw;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < g.KW && b + k >= 0 && b + k < g.Wi) { acc[k].x += t.x; acc[k].y += t.y; acc[k].z += t.z; acc[k].w += t.w; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[k][threadIdx.x][0] = acc[k].x; red[k][threadIdx.x][1] = acc[k].y;
        red[k][threadIdx.x][2] = acc[k].z; red[k][threadIdx.x][3] = acc[k].w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < g.KW * C; i += 256) {
        const int k = i / C, c = i % C;
        float s = 0.f;
        for (int j = 0; j < wlanes; ++j) s += red[k][j * cq + c / 4][c % 4];
        line_s[((size_t)line * g.KW + k) * C + c] = s;
    }
}

// S[tap][c] = sum over the lines (d', h') for which (kd, kh) read inside the grid, in fp64 and in a FIXED order: lane group
// lg of 16 adds lines lg, lg + 16, ... one after the other, then the 16 partial sums are added in index order (the order
// this kernel has always used -- results are bit-identical to the 27-workgroup form it replaces).
// One workgroup per (tap, 16 channels): 16 lane groups x 16 channels, up to 64 loads in flight per thread, the (d, h) of a line
// advanced by increments (a division per line was most of the kernel's instructions).
constexpr int kTapLanes = 16, kTapCh = 16, kTapDepth = 64;
__global__ void __launch_bounds__(kTapLanes * kTapCh)
k_tap_sums(ConvGeom g, const float* __restrict__ line_s, float* __restrict__ S) {
    __shared__ double red[kTapLanes][kTapCh];
    const int C = g.Cout;
    const int tap = blockIdx.x, kw = tap % g.KW, kh = (tap / g.KW) % g.KH, kd = tap / (g.KW * g.KH);
    const int nlines = g.Do * g.Ho;
    const int cl = threadIdx.x % kTapCh, lg = threadIdx.x / kTapCh;
    const int c = blockIdx.y * kTapCh + cl;
    double a = 0.0;
    if (c < C) {
        const float* src = line_s + (size_t)kw * C + c;
        const size_t ld = (size_t)g.KW * C;
        int d = lg / g.Ho, h = lg - d * g.Ho;                    // line = lg, then + kTapLanes per step
        const int dstep = kTapLanes / g.Ho, hstep = kTapLanes - dstep * g.Ho;
        auto inside = [&]() -> bool {
            const int bd = (d << g.ls_d) - g.pd + kd, bh = (h << g.ls_h) - g.ph + kh;
            return bd >= 0 && bd < g.Di && bh >= 0 && bh < g.Hi;
        };
        auto advance = [&]() {
            d += dstep; h += hstep;
            if (h >= g.Ho) { h -= g.Ho; ++d; }
        };
        // every load of a pass is issued before the first add: ONE memory round trip for the 800 lines of the Lyft grid (the
        // kernel runs beside a weight gradient that keeps HBM busy; each dependent round trip costs microseconds there)
        for (int line = lg; line < nlines; line += kTapDepth * kTapLanes) {
            float v[kTapDepth];
#pragma unroll
            for (int u = 0; u < kTapDepth; ++u) {
                const int l = line + u * kTapLanes;
                v[u] = src[(size_t)(l < nlines ? l : nlines - 1) * ld];      // (unconditional: clamped, not predicated)
            }
#pragma unroll
            for (int u = 0; u < kTapDepth; ++u) {
                if (line + u * kTapLanes < nlines) {
                    a += inside() ? (double)v[u] : 0.0;
                    advance();
                }
            }
        }
    }
    red[lg][cl] = a;
    __syncthreads();
    if (lg == 0 && c < C) {
        double t = red[0][cl];
        for (int l = 1; l < kTapLanes; ++l) t += red[l][cl];
        S[(size_t)tap * C + c] = (float)t;
    }
}

// g_all[c] = sum_tap sum_n W[tap][c][n] * S[tap][n]      (one block per c)
__global__ void __launch_bounds__(256)
k_const_field_gall(const float* __restrict__ W, const float* __restrict__ S, int ntaps, int Cin, int Cout,
                   float* __restrict__ g_all) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    double a = 0.0;
    for (int i = threadIdx.x; i < ntaps * Cout; i += 256) {
        const int tap = i / Cout, n = i - tap * Cout;
        a += (double)W[((size_t)tap * Cin + c) * Cout + n] * (double)S[i];
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) g_all[c] = (float)red[0];
}

// dW[tap][c][n] += cvec[c] * S[tap][n]
__global__ void k_const_field_dw(const float* __restrict__ S, const float* __restrict__ cvec,
                                 const int* __restrict__ cvec_row, int cvec_row_max, int ntaps, int Cin,
                                 int Cout, float* __restrict__ dW) {
    if (cvec_row) {                                   // the constant is row *cvec_row of a (rows, Cin) table
        int r = *cvec_row;
        if (r > cvec_row_max) r = cvec_row_max;
        cvec += (size_t)r * Cin;
    }
    const long long total = (long long)ntaps * Cin * Cout;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i % Cout);
        const long long t = i / Cout;
        const int c = (int)(t % Cin), tap = (int)(t / Cin);
        dW[i] = fmaf(cvec[c], S[(size_t)tap * Cout + n], dW[i]);
    }
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" size_t lisec_conv_tap_sums_workspace_bytes(const lisec_conv_geom* c) {
    ConvGeom g;
    if (conv_geom_check(c, &g)) return 0;
    return align_up(sizeof(float) * (size_t)g.Do * g.Ho * g.KW * g.Cout, 256);
}

extern "C" int lisec_conv_tap_sums_bn(const lisec_conv_geom* c, const float* dz, const float* y, const float* bnstate,
                                      const float* coef, float* dy, float* S, void* workspace, size_t workspace_bytes,
                                      lisec_stream_t stream_) {
    ConvGeom g;
    if (int rc = conv_geom_check(c, &g)) return rc;
    LISEC_CHECK_ARG(c->mode == 0 && dz && workspace, "tap sums: mode-0 geometry and non-NULL pointers required");
    LISEC_CHECK_ARG(g.Cout % 4 == 0 && g.Cout <= 256 && 256 % (g.Cout / 4) == 0 && g.out_stride % 4 == 0,
                    "tap sums: Cout/4 must divide 256");
    const bool apply = y != nullptr;
    LISEC_CHECK_ARG(!apply || (bnstate && coef && dy && g.out_stride == g.Cout &&
                               (((uintptr_t)y | (uintptr_t)bnstate | (uintptr_t)coef | (uintptr_t)dy) & 15) == 0),
                    "tap sums with the BatchNormalization apply pass: y, bnstate, coef, dy (16-byte aligned), dense rows");
    if (workspace_bytes < lisec_conv_tap_sums_workspace_bytes(c)) {
        set_error("tap sums workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    float* line_s = static_cast<float*>(workspace);
    if (apply)
        LISEC_LAUNCH(k_line_sums<true>, dim3(g.Do * g.Ho), dim3(256), 0, st, g, dz, line_s, y, bnstate, coef, dy);
    else
        LISEC_LAUNCH(k_line_sums<false>, dim3(g.Do * g.Ho), dim3(256), 0, st, g, dz, line_s, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
    if (S) LISEC_LAUNCH(k_tap_sums, dim3(g.KD * g.KH * g.KW, cdiv(g.Cout, kTapCh)), dim3(kTapLanes * kTapCh), 0, st, g, line_s, S);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_conv_tap_sums_finish(const lisec_conv_geom* c, const void* workspace, size_t workspace_bytes, float* S,
                                          lisec_stream_t stream_) {
    ConvGeom g;
    if (int rc = conv_geom_check(c, &g)) return rc;
    LISEC_CHECK_ARG(c->mode == 0 && S && workspace && workspace_bytes >= lisec_conv_tap_sums_workspace_bytes(c),
                    "tap sums: mode-0 geometry, S and the line-sum workspace of lisec_conv_tap_sums_bn");
    LISEC_LAUNCH(k_tap_sums, dim3(g.KD * g.KH * g.KW, cdiv(g.Cout, kTapCh)), dim3(kTapLanes * kTapCh), 0,
                 static_cast<hipStream_t>(stream_), g, static_cast<const float*>(workspace), S);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_conv_tap_sums(const lisec_conv_geom* c, const float* dy, float* S, void* workspace,
                                   size_t workspace_bytes, lisec_stream_t stream_) {
    return lisec_conv_tap_sums_bn(c, dy, nullptr, nullptr, nullptr, nullptr, S, workspace, workspace_bytes, stream_);
}

extern "C" int lisec_const_field_grads(const float* W, const float* S, const float* cvec, const int32_t* cvec_row,
                                       int cvec_row_max, int ntaps, int Cin, int Cout, float* dW, float* g_all,
                                       lisec_stream_t stream_) {
    LISEC_CHECK_ARG(S && ntaps > 0 && Cin > 0 && Cout > 0, "bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (g_all) {
        LISEC_CHECK_ARG(W, "W is needed for g_all");
        LISEC_LAUNCH(k_const_field_gall, dim3(Cin), dim3(256), 0, st, W, S, ntaps, Cin, Cout, g_all);
    }
    if (dW) {
        LISEC_CHECK_ARG(cvec, "cvec is needed for dW");
        long long total = (long long)ntaps * Cin * Cout;
        int gb = cdiv(total, 256);
        if (gb > 2048) gb = 2048;
        LISEC_LAUNCH(k_const_field_dw, dim3(gb), dim3(256), 0, st, S, cvec, cvec_row, cvec_row_max, ntaps, Cin, Cout, dW);
    }
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
