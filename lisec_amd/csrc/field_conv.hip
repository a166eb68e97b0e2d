// Forward of the FIRST Conv3D over the VFE's compact output (gfx950): a constant field plus V voxel rows.
//
// The tensor that layer reads (model_training.py:235-236) holds one constant vector c on every empty cell and a
// per-voxel value c + delta_v on the V occupied ones (vfe.hip: vout / delta).  The convolution is linear, so
//     y[p] = bias + sum_{tap reads inside the grid} W[tap]^T c            (depends on p only through the boundary)
//                 + sum_{tap reads an occupied cell v} W[tap]^T delta_v
// and the 70.8 GFLOP dense contraction over 640 000 cells (98.5 % of them empty on a Lyft sweep) becomes
//   1. k_field_taps     Z[tap][v] = W[tap]^T delta_v for the V voxel rows (+ Zc[tap] = W[tap]^T c): 128x64x64 MFMA tiles,
//                       one workgroup per (128 rows, (kd, kh)), skipped when no row of the tile feeds an output through
//                       that pair (depth stride 2: half of the (row, tap) pairs) -- ~1 GFLOP at 9 400 voxels;
//   2. k_field_combine  every output position: the boundary-class constant, plus -- where the 27-cell neighbourhood
//                       holds a voxel (cell_voxel map, staged per output line in LDS) -- its Z rows in tap order;
//                       writes y once (82 MB, HBM-bound) and feeds the BatchNormalization sums to a lisec_bn_sink.
// Same numbers as the dense kernel up to fp32 summation order, every sum in a fixed order (deterministic).  The dense
// grid is never formed: with the sparse backward of sparse_grid.hip neither direction of the training step needs it.
#include "conv.h"


namespace lisec {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FM = 128, FLDA = 68, FC = 64;          // rows per tile, padded LDS row, channels (in == out == 64)
constexpr int kFieldThreads = 256;
constexpr int kCombineThreads = 512;                 // k_field_combine: 32 output positions x 16 lanes per pass

// diagnostic: 100 MHz s_memrealtime stamps of thread 0 of every workgroup (tools/field_stamps.py), nullptr = off
__device__ unsigned long long* g_field_stamps = nullptr;
#define FIELD_STAMP(KERNEL_, WG_, K_)                                                                         \
    do {                                                                                                      \
        if (stamps && threadIdx.x == 0 && (WG_) < 8192)                                                       \
            stamps[((size_t)(KERNEL_) * 8192 + (WG_)) * 8 + (K_)] = __builtin_amdgcn_s_memrealtime();         \
    } while (0)

// does input coordinate x feed an output through kernel index k?  (o*stride - pad + k == x for some 0 <= o < n_out)
__device__ __forceinline__ bool feeds(int x, int k, int ls, int pad, int n_out) {
    const int t = x + pad - k;
    return t >= 0 && (t & ((1 << ls) - 1)) == 0 && (t >> ls) < n_out;
}

// One workgroup per (128 voxel rows, tpw taps): the A tile is staged once and the taps run over it, the next tap's W slab
// prefetched into registers under the MFMAs of the current one.
__global__ void __launch_bounds__(kFieldThreads)
k_field_taps(ConvGeom g, const float* __restrict__ vout, const float* __restrict__ delta,
             const int* __restrict__ info, const int* __restrict__ coords, int cap,
             const float* __restrict__ wp, float* __restrict__ Z, long long zstride, float* __restrict__ Zc, int tpw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int any_valid;
    float* sA = smem;
    float* sB = smem + FM * FLDA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* stamps = g_field_stamps;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    FIELD_STAMP(0, wg, 0);
    int V = info[LISEC_VI_NVOX];
    if (V > cap) V = cap;
    const int m0 = blockIdx.x * FM;
    if (m0 > V) return;                                   // rows 0 .. V exist (row V = the empty-cell constant)
    FIELD_STAMP(0, wg, 1);
    // tpw taps per workgroup: KW (one (kd, kh) pair, the A tile staged once) or 1 (three times the workgroups)
    const int tap0 = blockIdx.y * tpw;
    const int kh = (tap0 / g.KW) % g.KH, kd = tap0 / (g.KW * g.KH), kw0 = tap0 % g.KW;
    if (tid == 0) any_valid = 0;
    __syncthreads();
    if (tid < FM) {
        const int m = m0 + tid;
        bool ok = m == V;                                 // the constant is needed for every tap
        if (m < V)
            ok = feeds(coords[3 * m], kd, g.ls_d, g.pd, g.Do) && feeds(coords[3 * m + 1], kh, g.ls_h, g.ph, g.Ho) &&
                 (tpw > 1 || feeds(coords[3 * m + 2], kw0, g.ls_w, g.pw, g.Wo));
        if (ok) any_valid = 1;
    }
    __syncthreads();
    if (!any_valid) return;                               // (voxels are sorted by cell: a tile mostly shares its depth)
    FIELD_STAMP(0, wg, 2);

    const int piece = tid & 15;
    float4 rb[4];
    auto load_w = [&](int tap) {
        const float* wt = wp + (size_t)tap * FC * FC;     // packed [k/4][n][4] slab of this tap
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const float4*>(wt + (i * 256 + tid) * 4);
    };
    load_w(tap0);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = p * 16 + (tid >> 4), m = m0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m <= V) v = *reinterpret_cast<const float4*>((m < V ? delta + (size_t)m * FC : vout + (size_t)V * FC) + piece * 4);
        *reinterpret_cast<float4*>(sA + r * FLDA + piece * 4) = v;
    }
    const float* aRow = sA + (wave * 32 + (lane & 31)) * FLDA + 4 * (lane >> 5);
    const float* bCol = sB + ((lane >> 5) * FC + (lane & 31)) * 4;
    const int col = lane & 31;
    for (int kw = 0; kw < tpw; ++kw) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(sB + (i * 256 + tid) * 4) = rb[i];
        __syncthreads();
        if (kw == 0) FIELD_STAMP(0, wg, 3);
        if (kw + 1 < tpw) load_w(tap0 + kw + 1);
        f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
        for (int kc = 0; kc < FC / 8; ++kc) {
            const float4 a = *reinterpret_cast<const float4*>(aRow + kc * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(bCol + kc * 2 * FC * 4);
            const float4 b1 = *reinterpret_cast<const float4*>(bCol + kc * 2 * FC * 4 + 32 * 4);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
        }
        const int tap = tap0 + kw;
        if (kw == 0) { asm volatile("s_nop 0" ::"v"(acc0[0]), "v"(acc1[15])); FIELD_STAMP(0, wg, 4); }
        float* zt = Z + (size_t)tap * zstride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m <= V) {
                // row V is the empty-cell constant: it goes to a fixed place (Zc[tap]) so that its readers need not know V
                float* dst = m < V ? zt + (size_t)m * FC : Zc + (size_t)tap * FC;
                dst[col] = acc0[r];
                dst[32 + col] = acc1[r];
            }
        }
        __syncthreads();                                   // the W slab is overwritten next
        if (kw == 0) FIELD_STAMP(0, wg, 5);
    }
    FIELD_STAMP(0, wg, 6);
}

// One workgroup per segment of an output line (d', h'); 16 lanes x float4 per output position.  Nothing in here depends
// on the device-side voxel count: the constant's row lives at a fixed place (Zc), ordinals are clamped to the capacity,
// so the cell lookups, the constant rows and the bias are all requested at once (one round trip before the first store).
__global__ void __launch_bounds__(kCombineThreads, 8)
k_field_combine(ConvGeom g, const int* __restrict__ cell_voxel, int cap, const float* __restrict__ Z,
                long long zstride, const float* __restrict__ Zc, const float* __restrict__ bias,
                float* __restrict__ out, int nseg, int seg_len, int LW) {
    extern __shared__ __attribute__((aligned(16))) int dyn[];
    __shared__ __attribute__((aligned(16))) float sT[3][FC];      // per kw: sum over the line's valid (kd, kh) of W[tap]^T c
    __shared__ __attribute__((aligned(16))) float sbase[8][FC];   // per kw-validity mask: bias + those sums
    __shared__ __attribute__((aligned(16))) float red[2][kCombineThreads / 16][FC];
    int* sidx = dyn;                                              // [KD*KH][LW] voxel ordinal of the staged cells (-1: none)
    unsigned* smask = reinterpret_cast<unsigned*>(dyn + g.KD * g.KH * LW);   // [seg_len] taps that read an occupied cell
    const int tid = threadIdx.x;
    unsigned long long* stamps = g_field_stamps;
    FIELD_STAMP(1, blockIdx.x, 0);
    const int line = blockIdx.x / nseg, seg = blockIdx.x - line * nseg;
    const int dq = line / g.Ho, hq = line - dq * g.Ho;
    const int w0 = seg * seg_len;
    const int wn = g.Wo - w0 < seg_len ? g.Wo - w0 : seg_len;
    const int nkk = g.KD * g.KH;
    unsigned jvalid = 0;
    int jline[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int kd = j / g.KH, kh = j - kd * g.KH;
        const int di = (dq << g.ls_d) - g.pd + kd, hi = (hq << g.ls_h) - g.ph + kh;
        const bool ok = j < nkk && di >= 0 && di < g.Di && hi >= 0 && hi < g.Hi;
        jvalid |= ok ? 1u << j : 0u;
        jline[j] = ok ? (di * g.Hi + hi) * g.Wi : 0;
    }
    // ---- stage the voxel ordinals of the input lines this segment reads (all loads first, then the LDS writes) ----
    const int span = (wn - 1) * (1 << g.ls_w) + g.KW;
    const int wi0 = (w0 << g.ls_w) - g.pw;
    for (int e0 = 0; e0 < span; e0 += kCombineThreads) {
        const int e = e0 + tid, wi = wi0 + e;
        const bool inw = e < span && wi >= 0 && wi < g.Wi;
        int v[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] = (inw && ((jvalid >> j) & 1)) ? cell_voxel[(size_t)jline[j] + wi] : -1;
        if (e < span) {
#pragma unroll
            for (int j = 0; j < 9; ++j)
                if (j < nkk) sidx[j * LW + e] = v[j] <= cap ? v[j] : -1;
        }
    }
    // ---- the constant part: per kw, summed over the valid (kd, kh) in index order -----------------------------
    if (tid < g.KW * FC) {
        const int kw = tid / FC, n = tid - kw * FC;
        float t[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) t[j] = ((jvalid >> j) & 1) ? Zc[(size_t)(j * g.KW + kw) * FC + n] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) s += t[j];
        sT[kw][n] = s;
    }
    __syncthreads();
    FIELD_STAMP(1, blockIdx.x, 1);
    for (int i = tid; i < 8 * FC; i += kCombineThreads) {
        const int m = i / FC, n = i - m * FC;
        float b = bias ? bias[n] : 0.f;
        for (int kw = 0; kw < g.KW; ++kw)
            if ((m >> kw) & 1) b += sT[kw][n];
        sbase[m][n] = b;
    }
    for (int p = tid; p < wn; p += kCombineThreads) {
        int v[27];                                                // all lookups in flight at once (LDS latency, not 27 round trips)
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) v[j * 3 + kw] = (j < nkk && kw < g.KW) ? sidx[j * LW + (p << g.ls_w) + kw] : -1;
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                if (v[j * 3 + kw] >= 0) m |= 1u << (j * g.KW + kw);
        smask[p] = m;
    }
    __syncthreads();
    FIELD_STAMP(1, blockIdx.x, 2);
    // ---- every output position of the segment, two per thread and pass (their first gathers overlap) -----------------
    const int q = tid & 15, pr = tid >> 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    float* oline = out + (size_t)line * g.Wo * g.out_stride;
    auto base_of = [&](int p) -> float4 {
        const int bw = ((w0 + p) << g.ls_w) - g.pw;
        int wm = 0;
        for (int kw = 0; kw < g.KW; ++kw) wm |= (bw + kw >= 0 && bw + kw < g.Wi) ? (1 << kw) : 0;
        return *reinterpret_cast<const float4*>(&sbase[wm][q * 4]);
    };
    auto zrow = [&](int p, int t) -> const float4* {
        const int j = t / g.KW, kw = t - j * g.KW;
        const int v = sidx[j * LW + (p << g.ls_w) + kw];
        return reinterpret_cast<const float4*>(Z + (size_t)t * zstride + (size_t)v * FC + q * 4);
    };
    auto finish = [&](int p, float4 val, unsigned m) {
        while (m) {
            const int t = __ffs(m) - 1;
            m &= m - 1;
            const float4 z = *zrow(p, t);
            val.x += z.x; val.y += z.y; val.z += z.z; val.w += z.w;
        }
        *reinterpret_cast<float4*>(oline + (size_t)(w0 + p) * g.out_stride + q * 4) = val;
        s1.x += val.x; s1.y += val.y; s1.z += val.z; s1.w += val.w;
        s2.x = fmaf(val.x, val.x, s2.x); s2.y = fmaf(val.y, val.y, s2.y);
        s2.z = fmaf(val.z, val.z, s2.z); s2.w = fmaf(val.w, val.w, s2.w);
    };
    constexpr int kPass = kCombineThreads / 16;                   // positions per pass and half
    for (int p = pr; p < wn; p += 2 * kPass) {
        const int p2 = p + kPass;
        const bool has2 = p2 < wn;
        float4 va = base_of(p), vb = has2 ? base_of(p2) : va;
        unsigned ma = smask[p], mb = has2 ? smask[p2] : 0u;
        // first contribution of both positions in flight together (fixed order within each position)
        float4 za = make_float4(0.f, 0.f, 0.f, 0.f), zb = za;
        if (ma) { const int t = __ffs(ma) - 1; ma &= ma - 1; za = *zrow(p, t); }
        if (mb) { const int t = __ffs(mb) - 1; mb &= mb - 1; zb = *zrow(p2, t); }
        va.x += za.x; va.y += za.y; va.z += za.z; va.w += za.w;
        vb.x += zb.x; vb.y += zb.y; vb.z += zb.z; vb.w += zb.w;
        finish(p, va, ma);
        if (has2) finish(p2, vb, mb);
    }
    FIELD_STAMP(1, blockIdx.x, 3);
    if (!g.sink.acc) return;
    *reinterpret_cast<float4*>(&red[0][pr][q * 4]) = s1;
    *reinterpret_cast<float4*>(&red[1][pr][q * 4]) = s2;
    __syncthreads();
    if (tid < 2 * FC) {
        const int which = tid >> 6, n = tid & 63;
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < kCombineThreads / 16; ++k) v += (double)red[which][k][n];
        sink_add(g.sink, which, n, v);
    }
    FIELD_STAMP(1, blockIdx.x, 4);
    sink_finish(g.sink);
    FIELD_STAMP(1, blockIdx.x, 5);
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" size_t lisec_conv_field_forward_workspace_bytes(const lisec_conv_geom* c, int row_capacity) {
    if (!c || row_capacity < 0) return 0;
    // Z[tap][row_capacity + 1][64] (rows up to the capacity stay addressable) + Zc[tap][64]
    return align_up(sizeof(float) * (size_t)c->KD * c->KH * c->KW * ((size_t)row_capacity + 2) * FC, 256);
}

extern "C" int lisec_conv_field_forward(const lisec_conv_geom* c, const float* vout, const float* delta,
                                        const int32_t* info, const int32_t* coords, const int32_t* cell_voxel,
                                        int row_capacity, const float* packed_w, const float* bias, float* out,
                                        const lisec_bn_sink* sk, void* workspace, size_t workspace_bytes,
                                        lisec_stream_t stream_) {
    ConvGeom g;
    if (int rc = conv_geom_check(c, &g)) return rc;
    LISEC_CHECK_ARG(c->mode == 0 && !c->ps && g.Cin == FC && g.Cout == FC && g.CoutP == FC && g.in_stride == FC,
                    "field conv: a mode-0 contraction with 64 input and 64 output channels");
    LISEC_CHECK_ARG(g.KD <= 3 && g.KH <= 3 && g.KW <= 3, "field conv: at most 3 taps per axis");
    LISEC_CHECK_ARG(vout && delta && info && coords && cell_voxel && packed_w && out && workspace && row_capacity >= 0,
                    "field conv: NULL pointer");
    LISEC_CHECK_ARG((((uintptr_t)vout | (uintptr_t)delta | (uintptr_t)packed_w | (uintptr_t)out | (uintptr_t)workspace) & 15) == 0 &&
                    g.out_stride % 4 == 0, "field conv: 16-byte aligned tensors");
    if (workspace_bytes < lisec_conv_field_forward_workspace_bytes(c, row_capacity)) {
        set_error("field conv workspace too small");
        return LISEC_ENOSPC;
    }
    if (sk) {
        LISEC_CHECK_ARG(sk->acc && sk->n_rows > 0 && sk->kind == LISEC_SINK_FORWARD && sk->gamma && sk->beta && sk->bnstate &&
                        (sk->moving_mean == nullptr) == (sk->moving_var == nullptr),
                        "field conv: a forward bn sink needs accumulators, gamma/beta/bnstate");
    }
    // whole lines for small sweeps (800 workgroups feed the sink instead of 3200: 23 -> 6 us of atomics); big sweeps
    // have crowded lines next to empty ones, there two segments per line balance the pass (84 000 voxels: 328 -> 270 us)
    const int seg_env = tuning().field_seg >= 16 && tuning().field_seg <= 1024 ? tuning().field_seg : 0;
    const int seg_target = seg_env ? seg_env : (row_capacity > 65536 ? 256 : 512);
    const int nseg = cdiv(g.Wo, seg_target), seg_len = cdiv(g.Wo, nseg);
    const int LW = (seg_len - 1) * (1 << g.ls_w) + g.KW;
    const int nblocks = g.Do * g.Ho * nseg;
    if (sk) {
        g.sink.acc = static_cast<long long*>(sk->acc);
        g.sink.kind = sk->kind; g.sink.C = g.Cout; g.sink.unbiased = sk->unbiased_moving;
        g.sink.total = (unsigned)nblocks;
        g.sink.N = sk->n_rows;
        g.sink.gamma = sk->gamma; g.sink.beta = sk->beta; g.sink.mmean = sk->moving_mean; g.sink.mvar = sk->moving_var;
        g.sink.bnstate = sk->bnstate; g.sink.dgamma = nullptr; g.sink.dbeta = nullptr; g.sink.coef = nullptr;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    float* Z = static_cast<float*>(workspace);
    const long long zstride = ((long long)row_capacity + 1) * FC;
    const int ntaps = g.KD * g.KH * g.KW;
    float* Zc = Z + (size_t)ntaps * zstride;
    // small sweeps: one tap per workgroup (9 400 voxels: 25 us against 38 us -- the launch is far from filling the chip);
    // big ones: the KW taps of a (kd, kh) pair share one staged A tile (84 000 voxels: 268 us against 302 us for the call)
    const int tpw_env = tuning().field_tpw;
    const int tpw = (tpw_env == 1 || tpw_env == g.KW) ? tpw_env : (row_capacity > 65536 ? g.KW : 1);
    LISEC_LAUNCH(k_field_taps, dim3(cdiv((long long)row_capacity + 1, FM), ntaps / tpw), dim3(kFieldThreads),
                       (size_t)(FM * FLDA + FC * FC) * sizeof(float), st, g, vout, delta, info, coords, row_capacity,
                       packed_w, Z, zstride, Zc, tpw);
    const size_t dyn = sizeof(int) * ((size_t)g.KD * g.KH * LW + seg_len);
    LISEC_LAUNCH(k_field_combine, dim3(nblocks), dim3(kCombineThreads), dyn, st, g, cell_voxel, row_capacity,
                       (const float*)Z, zstride, (const float*)Zc, bias, out, nseg, seg_len, LW);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

// Diagnostic (not in lisec_hip.h): points the field kernels' stamp buffer at `buf` (device, 2*8192*8 uint64) or NULL.
extern "C" int lisec_debug_field_stamps(unsigned long long* buf) {
    LISEC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_field_stamps), &buf, sizeof(buf)));
    return LISEC_OK;
}
