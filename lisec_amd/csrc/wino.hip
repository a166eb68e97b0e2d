// Winograd F(2x2, 3x3) form of the stride-1 3x3 (h, w) contractions on the fp32 matrix cores of gfx950.
//
// The middle Conv3D blocks (model_training.py:193, 237-238), the stride-1 Conv2Ds of the first RPN block (:210-214) and their
// data gradients run at the POWER ceiling of v_mfma_f32_32x32x2_f32 (~122 TFLOP/s executed): what is
// left is fewer MFMAs per output.  For a 3x3 stride-1 pad-1 correlation a 2x2 block of outputs is
//     Y = A^T [ sum_c (G g_c G^T) . (B^T d_c B) ] A          (Lavin & Gray; d: 4x4 input patch, g: 3x3 kernel)
// 16 products per channel pair instead of 36: the layer becomes 16 independent contractions (one per transform point)
//     M_pt[tile, n] = sum_{kd} sum_c V_pt[tile, (kd, c)] * U_pt[(kd, c), n]
// over tiles x (depth taps x Cin) x Cout -- 4 / 9 of the MFMAs of the direct form.  fp32 in / fp32 accumulate throughout;
// B^T and A^T hold only 0 and +-1 (exact adds), G holds +-1/2 (exact scalings): the result differs from the fmaf chain of the
// direct kernels by summation order and by the transforms' own roundings (measured in tests/test_gpu_winograd.py against
// the fp64 oracle, same bound as the direct kernels).
//
// One 512-thread workgroup per CU owns 8 x 8 Winograd tiles (16 x 16 outputs) of one output plane x 64 output channels.
// Wave (half, wm, wn) owns 32 tiles x 32 channels x 8 of the 16 points (half = rows i of M: 128 accumulator registers per
// lane, two waves per SIMD); waves 0-3 also do ALL of the staging.  K runs in chunks of 8 input channels of one depth tap;
// per chunk
//   window          the raw 18 x 18 positions under the block, 16 channels of one depth plane at a time (two chunks), parked
//                   in LDS: every position is fetched ONCE, as 64 contiguous bytes
//   V[pt][tile][8]  each staging thread reads the 4 x 4 patch of one (tile, channel pair) from the window, gates it (the
//                   BatchNormalization(+ReLU) of the producing layer is applied here, padding stays exactly zero), transforms
//                   it (32 adds per channel) and stores 16 x 8 bytes
//   U[pt][n][8]     linear 32 KB copy of the pre-transformed kernel (lisec_conv_pack_weights_winograd writes the LDS image,
//                   swizzle included)
//   32 MFMAs per wave: per point one ds_read_b128 of V and one of U (four K steps each).
// Two LDS stages + the window (153 KB): the image of chunk c + 1 is built while the MFMAs of chunk c read the other stage.
// Rows of V / U are 32 bytes; the 16-byte half a lane reads is swizzled by bit 3 of the row so that every 16-lane group of a
// ds_read_b128 covers all 64 banks.  The output transform is linear in the rows of M: each half forms its share of both
// output lines, keeps one and hands the other to its partner through LDS.  What was measured on the way (and why the form
// stops at 1.4-1.6x of the direct kernels instead of 2.25x): DESIGN.md section 4.3, profiles/r04_winograd.txt.
#include <type_traits>

#include "conv.h"

namespace lisec {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWinoThreads = 512;
constexpr int WT = 64;                      // Winograd tiles per workgroup (8 x 8)
constexpr int WN = 64;                      // output channels per workgroup
constexpr int WK = 8;                       // input channels per K chunk
constexpr int V_FLOATS = 16 * WT * WK;      // 8192
constexpr int U_FLOATS = 16 * WN * WK;      // 8192
constexpr int STAGE_FLOATS = V_FLOATS + U_FLOATS;
constexpr int R_POS = 18 * 18;                // positions of the raw window under 8 x 8 tiles
constexpr int R_STRIDE = 20;                  // floats per position in LDS: 16 channels + 4 pad
constexpr int R_ITEMS = 6;                    // 16-byte pieces per thread: 324 x 4 = 1296 <= 6 x 256
constexpr size_t kWinoLds = (2 * STAGE_FLOATS + R_POS * R_STRIDE) * sizeof(float);      // 156 992

// float index of element k (0..7) of row `row` (0..63) inside one point's [64][8] plane
__host__ __device__ __forceinline__ int wino_swz(int row, int k) {
    return row * 8 + ((((k >> 2) ^ ((row >> 3) & 1))) << 2) + (k & 3);
}

// diagnostic (tools/wino_stamps.py): per workgroup, thread 0: [0] start, [1] first image in LDS, [2] after the K loop, [3] end
// (100 MHz s_memrealtime), [4] / [5] s_memtime (shader cycles) at [1] / [2], [6] chunks; nullptr (the default) = no stamp executes
__device__ unsigned long long* g_wino_stamps = nullptr;
#define WINO_STAMP(K_, V_)                                                                       \
    do {                                                                                         \
        if (stamps && threadIdx.x == 0 && stamp_wg < 8192) stamps[(size_t)stamp_wg * 8 + (K_)] = (V_); \
    } while (0)

// EXP: timing-only variants (wrong results; tools/wino_stamps.py): 1 = no side work in the K loop, 2 = no global loads in it,
// 3 = no LDS stores in it, 4 = no MFMAs in the helper waves, 5 = no MFMAs in the staging waves, 6 = the chunk walks stand still
// TAG only changes the symbol name: bench.py launches the second middle block's forward through k_wino<false, 0, 1> so that
// its row in a rocprofv3 --stats summary is that layer alone (same code as TAG 0).
template <bool XF, int EXP = 0, int TAG = 0>
__global__ void __launch_bounds__(kWinoThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_wino(ConvGeom g, int mode, const float* __restrict__ in, const float* __restrict__ U,
       const float* __restrict__ bias, const float* __restrict__ in_bn, int flags, float* __restrict__ out,
       int BH, int BW, unsigned plane_list) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves 0-3 (half 0) stage AND multiply, waves 4-7 (half 1) only multiply: wave w and wave w + 4 sit on one SIMD and own
    // the same 32 tiles x 32 channels, points 0-7 and 8-15
    const int half = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int tid = threadIdx.x & 255;               // staging identity (half 0 only)
    const int H = g.Ho, W = g.Wo;
    unsigned long long* stamps = g_wino_stamps;
    const unsigned stamp_wg = blockIdx.y * gridDim.x + blockIdx.x;
    WINO_STAMP(0, __builtin_amdgcn_s_memrealtime());
    // ---- which block ------------------------------------------------------------------------------------------
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int per_plane = BH * BW;
    // plane_list: the output planes this launch covers, one nibble each.  The planes of a strided / transposed Conv3D run
    // different numbers of depth taps (the data gradient of the second middle block: 1, 2, 2, 1) and the dispatcher waits for
    // a slot on the CU whose turn it is: in one launch the unequal workgroups left a third of the chip idle (mean 175 of 256
    // workgroups alive), so the host issues one launch per tap count.
    const int dslot = blk / per_plane;
    const int dplane = (plane_list >> (4 * dslot)) & 15;
    const int brem = blk - dslot * per_plane;
    const int by = brem / BW, bx = brem - by * BW;
    const int nb = blockIdx.y;
    // ---- depth taps of this output plane (wave-uniform): bit kd of dmask = tap kd reads a plane inside the tensor -------------
    int dmask = 0, npairs = 0;
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
        bool ok = kd < g.KD;
        src_coord(dplane, kd, g.ls_d, g.pd, g.Di, mode, ok);
        dmask |= ok ? (1 << kd) : 0;
        npairs += ok ? 1 : 0;
    }
    const int ncc = g.Cin / WK;
    const int nchunks = (flags & 0x4000) ? 0 : npairs * ncc;
    // ---- the (tile, channel pair) this thread transforms -------------------------------------------------------------
    const int tl = tid >> 2, cp = tid & 3;
    const int ty = tl >> 3, tx = tl & 7;
    const int y0 = 2 * (by * 8 + ty) - 1, x0 = 2 * (bx * 8 + tx) - 1;
    unsigned pmask = 0;                              // bit 4 i + j: patch position (i, j) lies inside the map
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            pmask |= ((unsigned)(y0 + i) < (unsigned)H && (unsigned)(x0 + j) < (unsigned)W) ? (1u << (4 * i + j)) : 0u;
    const int vdst = tl * 8 + (((cp >> 1) ^ ((tl >> 3) & 1)) << 2) + (cp & 1) * 2;      // + pt * 512
    // ---- the raw window: the 18 x 18 positions under the block's 8 x 8 tiles, 16 channels of one depth plane at a time ----
    // Every position is fetched ONCE, as 64 contiguous bytes (four lanes x 16 B), and parked in LDS ([position][16 + 4 pad]
    // floats: the 8-byte patch reads of 8 tiles x 4 channel pairs then cover all 64 banks); the transform reads its 4 x 4
    // patches from there.  (First version: every thread loaded its own patch from global memory, 8 bytes per position -- 16
    // cache lines per wave instruction, every position fetched by up to four tiles: the loop was bound by the address unit,
    // ~1500 of 6200 cycles per chunk.)
    float* sR = smem + 2 * STAGE_FLOATS;
    unsigned rg_off[R_ITEMS];                        // BYTE offsets inside one plane (< 2^32: checked by the host), so that the
                                                     // loads take the scalar-base + 32-bit lane offset form (no 64-bit lane addresses)
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) {
        int item = tid + 256 * r;
        item = item < R_POS * 4 ? item : R_POS * 4 - 1;          // (the surplus threads repeat the last item: same bytes)
        const int pos = item >> 2, quad = item & 3;
        const int py = pos / 18, px = pos - py * 18;
        int y = 16 * by - 1 + py, x = 16 * bx - 1 + px;
        y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);                 // positions outside the map read a valid address: the
        x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);                 // transform gates them to zero (pmask)
        rg_off[r] = (flags & 0x1000) ? 0u : (unsigned)((y * W + x) * g.in_stride + quad * 4) * 4u;
    }
    // LDS offset of item tid + 256 r: (position) * R_STRIDE + quad * 4 = rl0 + 64 R_STRIDE r; the last round is partial
    const int rl0 = (tid >> 2) * R_STRIDE + (tid & 3) * 4;
    const int rl5 = tid + 256 * 5 < R_POS * 4 ? rl0 + 5 * 64 * R_STRIDE : (R_POS - 1) * R_STRIDE + 12;
    const int rsrc = ((2 * ty) * 18 + 2 * tx) * R_STRIDE + cp * 2;      // patch (0, 0) of this thread's tile in the window
    const size_t plane_floats = (size_t)H * W * g.in_stride;
    const int u_chunk = (flags & 0x2000) ? 0 : 16 * WN * WK;   // floats of one (depth tap, channel chunk, column block) image
    const int nnb = gridDim.y;
    const float relu_lo = (flags & LISEC_CONV_IN_RELU) ? 0.f : -INFINITY;

    // Walks over the chunk list (depth tap, channel chunk), each a fixed distance ahead of the MFMAs; past the last chunk a
    // walk stays there (the loop body is branch-free so that it can be interleaved with the MFMAs: the surplus images are
    // never read).  u_: the U image (two chunks ahead), x_: the on-load constants (two ahead), w_: the raw window (by
    // 16-channel groups = two chunks).
    const int kd0 = __builtin_ctz(dmask | 16);
    int u_kd = kd0, u_cc = 0, u_left = nchunks - 1;
    int w_kd = kd0, w_cc = 0, w_left = nchunks / 2 - 1;
    auto advance = [&](int& kd, int& cc, int& left, int step) {          // (selects, no branch)
        const bool more = left > 0;
        const bool wrap = cc + step == ncc;
        const int kd_next = __builtin_ctz((dmask >> (kd + 1)) | 16) + kd + 1;
        left -= more ? 1 : 0;
        kd = (more && wrap) ? kd_next : kd;
        cc = more ? (wrap ? 0 : cc + step) : cc;
    };
    // Everything that goes through registers on its way to LDS is loaded AND stored inside one chunk (no load is in flight at
    // a chunk boundary: the compiler waits for vmcnt(0) at the top of the loop body -- a wait count cannot be carried round the
    // loop edge -- and a load issued just before it would be waited for in full):
    float4 rw0, rw1, rw2;                            // half of the window's next 16 channels at a time (even chunks)
    float4 ub0, ub1, ub2, ub3;                       // half of the next U image at a time (named: an array stayed in scratch)
    float2 tsc = make_float2(1.f, 1.f), tsh = make_float2(0.f, 0.f);
    auto window_src = [&]() -> const float* {
        bool dok = true;
        const int sd = src_coord(dplane, w_kd, g.ls_d, g.pd, g.Di, mode, dok);
        return in + (size_t)sd * plane_floats + w_cc * WK;
    };
#define WINO_R_LD(R_, I_) { const u32x4 w_ = __builtin_amdgcn_raw_buffer_load_b128(wrs, rg_off[I_], 0, 0); \
                           R_ = make_float4(__uint_as_float(w_.x), __uint_as_float(w_.y), __uint_as_float(w_.z), __uint_as_float(w_.w)); }
#define WINO_R_ST(R_, I_) *reinterpret_cast<float4*>(sR + ((I_) < 5 ? rl0 + (I_) * 64 * R_STRIDE : rl5)) = R_
#define WINO_U_LD1(R_, Q_) { const u32x4 u_ = __builtin_amdgcn_raw_buffer_load_b128(urs, tid * 16, (Q_) * 4096, 0); \
                             R_ = make_float4(__uint_as_float(u_.x), __uint_as_float(u_.y), __uint_as_float(u_.z), __uint_as_float(u_.w)); }
#define WINO_U_LD4(Q_) { WINO_U_LD1(ub0, (Q_) + 0) WINO_U_LD1(ub1, (Q_) + 1) WINO_U_LD1(ub2, (Q_) + 2) WINO_U_LD1(ub3, (Q_) + 3) }
#define WINO_U_ST2(A_, B_, Q_) { *reinterpret_cast<float4*>(up + (Q_) * 1024) = A_; *reinterpret_cast<float4*>(up + ((Q_) + 1) * 1024) = B_; }
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(XF ? in_bn : in), 0,
                                                                         XF ? 8 * g.Cin : 0, 0x00020000);
    auto load_x = [&](float2& sc, float2& sh, int xc) {      // on-load constants of channel chunk xc (scalar base + lane offset)
        if (XF) {
            const u32x2 a_ = __builtin_amdgcn_raw_buffer_load_b64(xrs, cp * 8, xc * (WK * 4), 0);
            const u32x2 b_ = __builtin_amdgcn_raw_buffer_load_b64(xrs, cp * 8, (g.Cin + xc * WK) * 4, 0);
            sc = make_float2(__uint_as_float(a_.x), __uint_as_float(a_.y));
            sh = make_float2(__uint_as_float(b_.x), __uint_as_float(b_.y));
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = (f32x16){0};

    const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31);
    const int aoff = half * 8 * (WT * WK) + arow * 8 + (((lane >> 5) ^ ((arow >> 3) & 1)) << 2);
    const int boff = V_FLOATS + half * 8 * (WN * WK) + brow * 8 + (((lane >> 5) ^ ((brow >> 3) & 1)) << 2);

    float2 d[16], t[16];
    // rows i0, i0 + 1 of the thread's 4 x 4 patch of the window's channel half `ch_half` -> d (raw; gated two slices later)
    auto read_rows = [&](int i0, int ch_half) {
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[4 * i + j] = *reinterpret_cast<const float2*>(sR + rsrc + (i * 18 + j) * R_STRIDE + ch_half * WK);
    };
    // gate (+ BatchNormalization, ReLU) of patch row i, in place
    // (gate = false: a block whose whole window lies inside the map -- most of them -- has nothing to gate)
    auto gate_row = [&](int i, const float2& sc, const float2& sh, bool gate) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 4 * i + j;
            const bool ok = !gate || ((pmask >> e) & 1);
            float2 v = d[e];
            if (XF) {
                v.x = fmaxf(fmaf(v.x, sc.x, sh.x), relu_lo);
                v.y = fmaxf(fmaf(v.y, sc.y, sh.y), relu_lo);
            }
            d[e].x = ok ? v.x : 0.f;
            d[e].y = ok ? v.y : 0.f;
        }
    };
    // the block's 18 x 18 window lies inside the map (wave-uniform)
    const bool interior = 16 * by >= 1 && 16 * by + 16 <= H - 1 && 16 * bx >= 1 && 16 * bx + 16 <= W - 1;
    auto bt_col = [&](int j) {                       // B^T d, column j
        t[0 + j].x = d[0 + j].x - d[8 + j].x;   t[0 + j].y = d[0 + j].y - d[8 + j].y;
        t[4 + j].x = d[4 + j].x + d[8 + j].x;   t[4 + j].y = d[4 + j].y + d[8 + j].y;
        t[8 + j].x = d[8 + j].x - d[4 + j].x;   t[8 + j].y = d[8 + j].y - d[4 + j].y;
        t[12 + j].x = d[4 + j].x - d[12 + j].x; t[12 + j].y = d[4 + j].y - d[12 + j].y;
    };
    auto b_row = [&](int i, float* vp, bool store) { // (B^T d) B, row i, and its four 8-byte stores
        float2 v0, v1, v2, v3;
        v0.x = t[4 * i + 0].x - t[4 * i + 2].x; v0.y = t[4 * i + 0].y - t[4 * i + 2].y;
        v1.x = t[4 * i + 1].x + t[4 * i + 2].x; v1.y = t[4 * i + 1].y + t[4 * i + 2].y;
        v2.x = t[4 * i + 2].x - t[4 * i + 1].x; v2.y = t[4 * i + 2].y - t[4 * i + 1].y;
        v3.x = t[4 * i + 1].x - t[4 * i + 3].x; v3.y = t[4 * i + 1].y - t[4 * i + 3].y;
        if (store) {
            *reinterpret_cast<float2*>(vp + (4 * i + 0) * (WT * WK)) = v0;
            *reinterpret_cast<float2*>(vp + (4 * i + 1) * (WT * WK)) = v1;
            *reinterpret_cast<float2*>(vp + (4 * i + 2) * (WT * WK)) = v2;
            *reinterpret_cast<float2*>(vp + (4 * i + 3) * (WT * WK)) = v3;
        } else {                                     // (timing variant: keeps the transform alive without the stores)
            asm volatile("" :: "v"(v0.x), "v"(v0.y), "v"(v1.x), "v"(v1.y), "v"(v2.x), "v"(v2.y), "v"(v3.x), "v"(v3.y));
        }
    };

    if (nchunks > 0) {
        // window of group 0 -> LDS, image of chunk 0 -> stage 0 (the loop then prepares chunk c + 1 inside chunk c)
        if (half == 0) {
            const float* wsrc = window_src();
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsrc), 0, 0x7fffffff, 0x00020000);
            WINO_R_LD(rw0, 0); WINO_R_LD(rw1, 1); WINO_R_LD(rw2, 2);
            WINO_R_ST(rw0, 0); WINO_R_ST(rw1, 1); WINO_R_ST(rw2, 2);
            WINO_R_LD(rw0, 3); WINO_R_LD(rw1, 4); WINO_R_LD(rw2, 5);
            WINO_R_ST(rw0, 3); WINO_R_ST(rw1, 4); WINO_R_ST(rw2, 5);
            advance(w_kd, w_cc, w_left, 2);
            load_x(tsc, tsh, 0);
        }
        __syncthreads();
        if (half == 0) {
            const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(U + ((size_t)(u_kd * ncc + u_cc) * nnb + nb) * u_chunk), 0, 16 * WN * WK * 4, 0x00020000);
            float* up = smem + V_FLOATS + tid * 4;
            WINO_U_LD4(0)
            read_rows(0, 0);
            read_rows(2, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) gate_row(i, tsc, tsh, true);
#pragma unroll
            for (int j = 0; j < 4; ++j) bt_col(j);
#pragma unroll
            for (int i = 0; i < 4; ++i) b_row(i, smem + vdst, true);
            WINO_U_ST2(ub0, ub1, 0) WINO_U_ST2(ub2, ub3, 2)
            WINO_U_LD4(4)
            WINO_U_ST2(ub0, ub1, 4) WINO_U_ST2(ub2, ub3, 6)
            advance(u_kd, u_cc, u_left, 1);
            load_x(tsc, tsh, u_cc);                  // chunk 1's
        }
    }
    __syncthreads();
    WINO_STAMP(1, __builtin_amdgcn_s_memrealtime());
    WINO_STAMP(4, __builtin_amdgcn_s_memtime());
    // A wave's fp32 MFMA holds its issue slot for the whole 64 cycles: measured on the one-wave-per-SIMD form of this kernel,
    // the issue time of EVERY other instruction of the wave simply added to the MFMA time (4293 cycles per chunk with nothing
    // but MFMAs and fragment reads, 6039 with the staging: profiles/r04_winograd.txt) -- nothing hides in a lone wave.  So a
    // SIMD holds TWO waves: wave w (half 0) runs 32 MFMAs per chunk and ALL of the staging, wave w + 4 (half 1) runs the other
    // 32 MFMAs and nothing else; while the one stages, the other's MFMAs keep the pipe busy.  The compiler left alone puts the
    // whole transform in front of the MFMAs and the loads behind them, so the stager's chunk is cut by hand into 8 groups of
    // one transform point (two fragment reads, four MFMAs), each carrying two
    // slices p of the side work for chunk c + 1, pinned by sched_barrier:
    //   p 0-1    the thread's 4 x 4 patch: two rows (eight 8-byte reads from the window) per slice, gated two slices later --
    //            the side work alone is LATENCY-bound (4400 cycles per chunk with read-then-use slices and no MFMA at all)
    //   p 0, 8   four 16-byte loads of one half of the U image of chunk c + 1;  p 6-7, 14-15  their stores, two per slice
    //   p 1-2    EVEN chunks: three 16-byte loads each of the window's next 16 channels;  p 12-13  their stores.  The window
    //            is single: an even chunk reads its second half (p 0-1), ALL waves meet at a barrier after group 3 (p 7), and
    //            only then is it overwritten; the odd chunk that follows reads the new window after the barrier that ends
    //            the even one
    //   p 2-5    gate (+ BatchNormalization, ReLU) of patch row i;  p 6  the on-load constants of chunk c + 2
    //   p 6-9    B^T d (column j)
    //   p 10-13  (B^T d) B (row i) and its four 8-byte stores into the other stage
#define WINO_MFMA4(P_)                                                                                   \
    acc[P_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[P_], 0, 0, 0);                            \
    acc[P_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[P_], 0, 0, 0);                            \
    acc[P_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[P_], 0, 0, 0);                            \
    acc[P_] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[P_], 0, 0, 0);
    auto chunk = [&](auto par_tag, auto stager_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool STAGER = (decltype(stager_tag)::value & 1) != 0, GATE = (decltype(stager_tag)::value & 2) != 0;
        const float* st = smem + PAR * STAGE_FLOATS;
        float* nx = smem + (PAR ^ 1) * STAGE_FLOATS;
        const float* ap = st + aoff;
        const float* bp = st + boff;
        float4 a0 = *reinterpret_cast<const float4*>(ap);
        float4 b0 = *reinterpret_cast<const float4*>(bp);
        const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(U + ((size_t)(u_kd * ncc + u_cc) * nnb + nb) * u_chunk), 0, 16 * WN * WK * 4, 0x00020000);
        float* up = nx + V_FLOATS + tid * 4;
        const float* wsrc = STAGER ? window_src() : in;
        // (a buffer descriptor over the plane: scalar base + one 32-bit lane offset per load, no 64-bit lane addresses)
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsrc), 0, 0x7fffffff, 0x00020000);
        const int x_next = (u_left > 0) ? (u_cc + 1 == ncc ? 0 : u_cc + 1) : u_cc;      // channel chunk of chunk c + 2
        float* vp = nx + vdst;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            float4 a0n = a0, b0n = b0;
            if (pp + 1 < 8) {
                a0n = *reinterpret_cast<const float4*>(ap + (pp + 1) * (WT * WK));
                b0n = *reinterpret_cast<const float4*>(bp + (pp + 1) * (WN * WK));
            }
            if (STAGER && EXP != 1) {
#pragma unroll
                for (int p = 2 * pp; p < 2 * pp + 2; ++p) {
                    if (p == 0) WINO_U_LD4(0)
                    if (p == 0) read_rows(0, PAR ^ 1);           // chunk c + 1 is the OTHER half of the window's 16 channels
                    if (p == 1) read_rows(2, PAR ^ 1);
                    if (PAR == 0) {
                        if (p == 1) { WINO_R_LD(rw0, 0); WINO_R_LD(rw1, 1); WINO_R_LD(rw2, 2); }
                        if (p == 8) { WINO_R_ST(rw0, 0); WINO_R_ST(rw1, 1); WINO_R_ST(rw2, 2); }      // (behind the barrier)
                        if (p == 9) { WINO_R_LD(rw0, 3); WINO_R_LD(rw1, 4); WINO_R_LD(rw2, 5); }
                        if (p == 15) { WINO_R_ST(rw0, 3); WINO_R_ST(rw1, 4); WINO_R_ST(rw2, 5); }
                    }
                    if (p >= 2 && p < 6) gate_row(p - 2, tsc, tsh, GATE);       // (chunk c + 1's constants)
                    if (p == 6) load_x(tsc, tsh, x_next);                       // chunk c + 2's, once the gate is through
                    if (p >= 6 && p < 10) bt_col(p - 6);
                    if (p >= 10 && p < 14) b_row(p - 10, vp, true);
                    if (p == 6) WINO_U_ST2(ub0, ub1, 0)
                    if (p == 7) WINO_U_ST2(ub2, ub3, 2)
                    if (p == 8) WINO_U_LD4(4)
                    if (p == 14) WINO_U_ST2(ub0, ub1, 4)
                    if (p == 15) WINO_U_ST2(ub2, ub3, 6)
                }
            }
            if (!((EXP == 4 && !STAGER) || (EXP == 5 && STAGER))) { WINO_MFMA4(pp) }
            if (STAGER) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // VALU
                    __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);      // LDS read / write
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // global load
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PAR == 0 && pp == 3 && EXP != 1) __syncthreads();      // every wave has read the window's second half
            a0 = a0n; b0 = b0n;
        }
        if (STAGER && EXP != 6) {                    // (EXP 6: the walks stand still -- what the scalar work costs)
            advance(u_kd, u_cc, u_left, 1);
            if (PAR == 0) advance(w_kd, w_cc, w_left, 2);
        }
        __syncthreads();
    };
    if (half == 0 && interior) {
        for (int c = 0; c < nchunks; c += 2) {
            chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
    } else if (half == 0) {
        for (int c = 0; c < nchunks; c += 2) {
            chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
            chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
        }
    } else {
        for (int c = 0; c < nchunks; c += 2) {
            chunk(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            chunk(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        }
    }
#undef WINO_MFMA4
#undef WINO_R_LD
#undef WINO_R_ST
#undef WINO_U_LD4
#undef WINO_U_LD1
#undef WINO_U_ST2

    WINO_STAMP(2, __builtin_amdgcn_s_memrealtime());
    WINO_STAMP(5, __builtin_amdgcn_s_memtime());
    WINO_STAMP(6, (unsigned long long)nchunks);
    if (flags & 0x8000) { if (acc[3][5] == 123.f) out[0] = 1.f; return; }
    // ---- output transform A^T M A + the epilogue of the direct kernels --------------------------------------------------
    // accumulator register r of a lane: tile row (r >> 2) of the wave's four, tile column (r & 3) + 4 (lane >> 5), channel
    // lane & 31 -- the output line is wave-uniform (scalar row pointers), the position inside it one 32-bit lane offset.
    // A half holds two of the four rows i of M: it forms its share of both output lines a (the transform is linear), keeps
    // line a = half, hands line a = half ^ 1 to the wave it shares the tile with through LDS (the stages are dead), adds what
    // it is handed and stores its line.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                 // (keeps the epilogue's per-lane values out of the K loop's registers)
    const int col = lane_e & 31, hi = lane_e >> 5;
    const int n = nb * WN + wn * 32 + col;
    const bool nok = n < g.Cout;
    const float bv = (bias && nok) ? bias[n] : 0.f;
    const bool orelu = (flags & LISEC_CONV_OUT_RELU) != 0, accum = (flags & LISEC_CONV_ACCUMULATE) != 0;
    float ys = 1.f, yh = 0.f, ym = 0.f, yi = 0.f;
    if (g.bwd_y && nok) { ys = g.bwd_bn[n]; yh = g.bwd_bn[g.Cout + n]; ym = g.bwd_bn[2 * g.Cout + n]; yi = g.bwd_bn[3 * g.Cout + n]; }
    const int xt = 2 * (bx * 8 + 4 * hi);            // first output column of the lane's tile column 0
    const unsigned xo = (unsigned)(xt * g.out_stride + n), xy = (unsigned)(xt * g.Cout + n);
    // this half's share of output line a, columns b = 0, 1, from accumulator register r (A^T M A is linear in the rows of M)
    auto share = [&](int r, int a, float& ob0, float& ob1) {
        float s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m_lo = acc[j][r], m_hi = acc[4 + j][r];          // rows i = 2 half, 2 half + 1 of M
            s[j] = half == 0 ? (a == 0 ? m_lo + m_hi : m_hi) : (a == 0 ? m_lo : -m_lo - m_hi);
        }
        ob0 = s[0] + s[1] + s[2];
        ob1 = s[1] - s[2] - s[3];
    };
    {
        float* give = smem + (half ^ 1) * 8192 + ((wave & 3) * 32) * 64 + lane_e;      // [receiving half][tile wave][2 r + b][lane]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float g0, g1;
            share(r, half ^ 1, g0, g1);
            give[(2 * r) * 64] = g0;
            give[(2 * r + 1) * 64] = g1;
        }
    }
    __syncthreads();
    float sum = 0.f, sq = 0.f;
    // optional second contraction on the stored tile (lisec_conv_extras.tail_w, as in k_igemm_halo<..., TAIL>): the gated
    // 16 x 16 x 64 block also goes to LDS ([256 positions][LDT], behind the exchange scratch) as the tail's A operand
    const bool tail = g.tail_w != nullptr;
    constexpr int LDT = 68;
    float* tT = smem + 16384;
    // (full: every one of the block's 16 x 16 outputs and 64 channels exists -- no per-store predicate)
    auto store_lines = [&](auto full_tag) {
        constexpr bool full = decltype(full_tag)::value;
        const float* take = smem + half * 8192 + ((wave & 3) * 32) * 64 + lane_e;
        const int a2 = half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gty = by * 8 + wm * 4 + (r >> 2);
            const int y = 2 * gty + a2;
            const size_t line = ((size_t)dplane * H + y) * W;
            float* oline = out + line * g.out_stride;
            const float* mline = g.out_mask ? g.out_mask + line * g.out_stride : nullptr;
            const float* yline = (g.bwd_y && !tail) ? g.bwd_y + line * g.Cout : nullptr;
            float k0, k1;
            share(r, a2, k0, k1);
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int dx = 2 * (r & 3) + b2;
                const float part = take[(2 * r + b2) * 64], mine = b2 ? k1 : k0;
                if (full || (y < H && xt + dx < W && nok)) {
                    const unsigned off = xo + (unsigned)(dx * g.out_stride);
                    float v = (half == 0 ? mine + part : part + mine) + bv;       // (rows 0-1 of M first: one order)
                    if (accum) v += oline[off];
                    if (mline && !(mline[off] > 0.f)) v = 0.f;
                    if (orelu) v = fmaxf(v, 0.f);
                    oline[off] = v;
                    if (tail) {
                        tT[((2 * (wm * 4 + (r >> 2)) + a2) * 16 + 8 * hi + dx) * LDT + wn * 32 + col] = v;
                    } else if (yline) {
                        const float yv = yline[xy + (unsigned)(dx * g.Cout)];
                        const float dv = (g.bwd_relu && !(fmaf(yv, ys, yh) > 0.f)) ? 0.f : v;
                        sum += dv; sq = fmaf(dv, (yv - ym) * yi, sq);
                    } else {
                        sum += v; sq = fmaf(v, v, sq);
                    }
                } else if (tail) {                   // (outside the map: a zero row of the tail's operand)
                    tT[((2 * (wm * 4 + (r >> 2)) + a2) * 16 + 8 * hi + dx) * LDT + wn * 32 + col] = 0.f;
                }
            }
        }
    };
    if (16 * by + 16 <= H && 16 * bx + 16 <= W && nb * WN + WN <= g.Cout) store_lines(std::true_type{});
    else store_lines(std::false_type{});
    float sumB = 0.f, sqB = 0.f;                     // (tail: statistics of its second 32 columns)
    if (tail) {
        // Dense(64) backward riding on the block (model_training.py:195 backwards): dz = (gated block) @ Wd^T -> tail_out, with
        // the backward statistics of the BatchNormalization under the Dense.  Wave w owns the block's lines 2 w, 2 w + 1 (32
        // positions) x 64 columns; the B fragments come straight from the packed 64 x 64 kernel (16 KB, L2-resident).
        __syncthreads();
        f32x16 t0 = {0}, t1 = {0};
        const float* aRow = tT + (wave * 32 + (lane_e & 31)) * LDT + 4 * hi;
        const float* bC = g.tail_w + (hi * 64 + (lane_e & 31)) * 4;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            const float4 a = *reinterpret_cast<const float4*>(aRow + kc * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(bC + kc * 2 * 64 * 4);
            const float4 b1 = *reinterpret_cast<const float4*>(bC + kc * 2 * 64 * 4 + 32 * 4);
            t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, t1, 0, 0, 0);
        }
        const int nA = col, nB = 32 + col;           // the tail has 64 columns, row stride 64
        float sA = 1.f, hA = 0.f, mA = 0.f, iA = 0.f, sB = 1.f, hB = 0.f, mB = 0.f, iB = 0.f;
        if (g.bwd_y) {
            sA = g.bwd_bn[nA]; hA = g.bwd_bn[64 + nA]; mA = g.bwd_bn[128 + nA]; iA = g.bwd_bn[192 + nA];
            sB = g.bwd_bn[nB]; hB = g.bwd_bn[64 + nB]; mB = g.bwd_bn[128 + nB]; iB = g.bwd_bn[192 + nB];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;         // position 32 wave + row of the block: line, column
            const int y = 16 * by + 2 * wave + (row >> 4), x = 16 * bx + (row & 15);
            if (y < H && x < W) {
                const size_t pos = ((size_t)dplane * H + y) * W + x;
                const float va = t0[r], vb = t1[r];
                g.tail_out[pos * 64 + nA] = va;
                g.tail_out[pos * 64 + nB] = vb;
                if (g.bwd_y) {
                    const float ya = g.bwd_y[pos * 64 + nA], yb = g.bwd_y[pos * 64 + nB];
                    const float da = (g.bwd_relu && !(fmaf(ya, sA, hA) > 0.f)) ? 0.f : va;
                    const float db = (g.bwd_relu && !(fmaf(yb, sB, hB) > 0.f)) ? 0.f : vb;
                    sum += da; sq = fmaf(da, (ya - mA) * iA, sq);
                    sumB += db; sqB = fmaf(db, (yb - mB) * iB, sqB);
                } else {
                    sum += va; sq = fmaf(va, va, sq);
                    sumB += vb; sqB = fmaf(vb, vb, sqB);
                }
            }
        }
    }
    WINO_STAMP(3, __builtin_amdgcn_s_memrealtime());
    if (g.sink.acc && tail) {
        __syncthreads();
        float* red = smem;                           // [8 waves][4][32]
        sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
        sumB += __shfl_xor(sumB, 32, 64); sqB += __shfl_xor(sqB, 32, 64);
        if (lane < 32) {
            red[(wave * 4 + 0) * 32 + lane] = sum; red[(wave * 4 + 1) * 32 + lane] = sq;
            red[(wave * 4 + 2) * 32 + lane] = sumB; red[(wave * 4 + 3) * 32 + lane] = sqB;
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int q = threadIdx.x >> 5, c = threadIdx.x & 31;    // q: 0 sumA, 1 sqA, 2 sumB, 3 sqB
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += (double)red[(w * 4 + q) * 32 + c];
            sink_add(g.sink, q & 1, (q >> 1) * 32 + c, v);
        }
        sink_finish(g.sink);
    } else if (g.sink.acc) {
        __syncthreads();                             // (the exchange is over: the scratch below reuses its space)
        float* red = smem;                           // [8 waves][2][32]
        sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
        if (lane < 32) { red[(wave * 2 + 0) * 32 + lane] = sum; red[(wave * 2 + 1) * 32 + lane] = sq; }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = (threadIdx.x >> 5) & 1, hn = threadIdx.x >> 6, c = threadIdx.x & 31;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w)              // the four waves that hold column half hn: (half, wm) = (w >> 1, w & 1)
                v += (double)red[((((w >> 1) * 4 + (w & 1) * 2 + hn) * 2) + which) * 32 + c];
            const int ch = nb * WN + hn * 32 + c;
            if (ch < g.Cout) sink_add(g.sink, which, ch, v);
        }
        sink_finish(g.sink);
    }
}

// U image of one (depth tap, 8-channel chunk, 64-column block): [pt][n][8 k] with the kernel's swizzle; element (tap, k, n) of
// the source is read at src[tap * tap_stride + k * k_stride + n * n_stride], tap = (kd * 3 + kh) * 3 + kw.  flip: the (kh, kw)
// taps are mirrored (the data gradient of a stride-1 pad-1 correlation is the correlation with the mirrored kernel).
__global__ void k_wino_pack(const float* __restrict__ src, int KD, int K, int N, long long tap_stride, long long k_stride,
                            long long n_stride, int flip, int ncc, int nnb, float* __restrict__ dst) {
    const long long total = (long long)KD * ncc * 8 * nnb * 64;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int nn = (int)(i % 64);
        long long t = i / 64;
        const int kk = (int)(t % 8); t /= 8;
        const int b = (int)(t % nnb); t /= nnb;
        const int cc = (int)(t % ncc);
        const int kd = (int)(t / ncc);
        const int k = cc * 8 + kk, n = b * 64 + nn;
        float gk[3][3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int sh = flip ? 2 - kh : kh, sw = flip ? 2 - kw : kw;
                gk[kh][kw] = (k < K && n < N) ? src[((kd * 3 + sh) * 3 + sw) * tap_stride + k * k_stride + n * n_stride] : 0.f;
            }
        float tmp[4][3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            tmp[0][kw] = gk[0][kw];
            tmp[1][kw] = 0.5f * (gk[0][kw] + gk[1][kw] + gk[2][kw]);
            tmp[2][kw] = 0.5f * (gk[0][kw] - gk[1][kw] + gk[2][kw]);
            tmp[3][kw] = gk[2][kw];
        }
        float* d = dst + ((size_t)(kd * ncc + cc) * nnb + b) * (16 * WN * WK) + wino_swz(nn, kk);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            d[(4 * i4 + 0) * (WN * WK)] = tmp[i4][0];
            d[(4 * i4 + 1) * (WN * WK)] = 0.5f * (tmp[i4][0] + tmp[i4][1] + tmp[i4][2]);
            d[(4 * i4 + 2) * (WN * WK)] = 0.5f * (tmp[i4][0] - tmp[i4][1] + tmp[i4][2]);
            d[(4 * i4 + 3) * (WN * WK)] = tmp[i4][2];
        }
    }
}

// geometry + extras the Winograd kernel serves; msg: why not (for lisec_last_error)
bool wino_ok(const lisec_conv_geom* c, const ConvGeom& g, bool has_in_bn, int flags, const lisec_conv_extras* ex,
             const char** msg) {
#define LISEC_WINO_NEED(cond, text) do { if (!(cond)) { *msg = text; return false; } } while (0)
    LISEC_WINO_NEED(c->KH == 3 && c->KW == 3 && c->sh == 1 && c->sw == 1 && c->ph == 1 && c->pw == 1, "3x3 (h, w) taps, stride 1, pad 1");
    LISEC_WINO_NEED(c->Hi == c->Ho && c->Wi == c->Wo && c->Ho >= 2 && c->Wo >= 2, "equal input and output maps of at least 2 x 2");
    LISEC_WINO_NEED(!c->ps, "no pixel-shuffle store");
    LISEC_WINO_NEED(c->Cin % 16 == 0 && c->in_stride % 4 == 0, "Cin % 16 == 0 and in_stride % 4 == 0");
    LISEC_WINO_NEED((long long)c->Hi * c->Wi * c->in_stride < (1LL << 29), "one plane of the gathered tensor below 2 GB");
    LISEC_WINO_NEED(has_in_bn || !(flags & LISEC_CONV_IN_RELU), "LISEC_CONV_IN_RELU needs in_bnstate");
    LISEC_WINO_NEED(!(flags & LISEC_CONV_TAG_ROOFLINE) || !has_in_bn, "the roofline tag only without in_bnstate");
    if (ex) {
        LISEC_WINO_NEED(!ex->in_y && !ex->queue && !ex->dense_dw, "no backward on load, no row queue, no Dense weight gradient");
        LISEC_WINO_NEED(!ex->tail_w || (ex->tail_out && c->Cout == 64 && c->out_stride == 64 &&
                                        !(flags & (LISEC_CONV_ACCUMULATE | LISEC_CONV_OUT_RELU)) && (!ex->bwd_y || ex->sink) &&
                                        ((uintptr_t)ex->tail_w & 15) == 0),
                        "tail: a packed 64 x 64 kernel and an output, Cout = out_stride = 64, no accumulate / ReLU, statistics through a sink");
        LISEC_WINO_NEED(!ex->bwd_y || (ex->bwd_bnstate && ex->sink && ex->sink->kind == LISEC_SINK_BACKWARD),
                        "backward statistics go through a backward sink");
        LISEC_WINO_NEED(!ex->sink || ex->sink->kind == LISEC_SINK_BACKWARD || !ex->bwd_y, "a forward sink excludes bwd_y");
    }
#undef LISEC_WINO_NEED
    (void)g;
    return true;
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" size_t lisec_conv_winograd_packed_floats(int KD, int K, int N) {
    if (KD <= 0 || K <= 0 || N <= 0) return 0;
    return (size_t)KD * (align_up(K, 8) / 8) * (align_up(N, 64) / 64) * (16 * WN * WK);
}

extern "C" int lisec_conv_pack_weights_winograd(const float* src, int KD, int K, int N, long long tap_stride,
                                                long long k_stride, long long n_stride, int flip_hw, float* dst,
                                                lisec_stream_t stream_) {
    LISEC_CHECK_ARG(src && dst && KD > 0 && KD <= 4 && K > 0 && N > 0, "bad Winograd pack arguments");
    LISEC_CHECK_ARG(((uintptr_t)dst & 15) == 0, "dst must be 16-byte aligned");
    const int ncc = (int)(align_up(K, 8) / 8), nnb = (int)(align_up(N, 64) / 64);
    const long long total = (long long)KD * ncc * 8 * nnb * 64;
    int gb = cdiv(total, 256);
    if (gb > 8192) gb = 8192;
    LISEC_LAUNCH(k_wino_pack, dim3(gb), dim3(256), 0, static_cast<hipStream_t>(stream_), src, KD, K, N, tap_stride, k_stride,
                 n_stride, flip_hw ? 1 : 0, ncc, nnb, dst);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_conv_winograd_supported(const lisec_conv_geom* c, int has_in_bnstate, int flags,
                                             const lisec_conv_extras* extras) {
    ConvGeom g;
    if (conv_geom_check(c, &g)) return 0;
    const char* msg = "";
    return wino_ok(c, g, has_in_bnstate != 0, flags, extras, &msg) ? 1 : 0;
}

extern "C" int lisec_conv_forward_winograd(const lisec_conv_geom* c, const float* in, const float* wino_w, const float* bias,
                                           const float* in_bnstate, int flags, float* out, const lisec_conv_extras* extras,
                                           lisec_stream_t stream_) {
    ConvGeom g;
    if (int rc = conv_geom_check(c, &g)) return rc;
    const char* msg = "";
    LISEC_CHECK_ARG(wino_ok(c, g, in_bnstate != nullptr, flags, extras, &msg), "Winograd form needs: %s", msg);
    LISEC_CHECK_ARG(in && wino_w && out, "NULL tensor pointer");
    LISEC_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)wino_w & 15) == 0 && (!in_bnstate || ((uintptr_t)in_bnstate & 7) == 0),
                    "in and the Winograd kernel: 16-byte aligned");
    const int TH = (g.Ho + 1) / 2, TW = (g.Wo + 1) / 2;
    const int BH = cdiv(TH, 8), BW = cdiv(TW, 8);
    const int nnb = (int)(align_up(g.Cout, WN) / WN);
    if (extras) {
        g.out_mask = extras->out_mask;
        if (extras->tail_w) { g.tail_w = extras->tail_w; g.tail_out = extras->tail_out; }
        if (extras->bwd_y) { g.bwd_y = extras->bwd_y; g.bwd_bn = extras->bwd_bnstate; g.bwd_relu = extras->bwd_relu ? 1 : 0; }
        if (const lisec_bn_sink* sk = extras->sink) {
            LISEC_CHECK_ARG(sk->acc && sk->n_rows > 0, "bn sink: accumulators and a row count");
            LISEC_CHECK_ARG((sk->kind == LISEC_SINK_FORWARD && !extras->bwd_y && sk->gamma && sk->beta && sk->bnstate &&
                             (sk->moving_mean == nullptr) == (sk->moving_var == nullptr)) ||
                            (sk->kind == LISEC_SINK_BACKWARD && extras->bwd_y && sk->dgamma && sk->dbeta && sk->coef),
                            "bn sink: kind 1 needs gamma/beta/bnstate and no bwd_y, kind 2 needs bwd_y and dgamma/dbeta/coef");
            g.sink.acc = static_cast<long long*>(sk->acc);
            g.sink.kind = sk->kind; g.sink.C = g.Cout; g.sink.unbiased = sk->unbiased_moving;
            g.sink.total = (unsigned)(g.Do * BH * BW) * (unsigned)nnb;
            g.sink.N = sk->n_rows;
            g.sink.gamma = sk->gamma; g.sink.beta = sk->beta; g.sink.mmean = sk->moving_mean; g.sink.mvar = sk->moving_var;
            g.sink.bnstate = sk->bnstate; g.sink.dgamma = sk->dgamma; g.sink.dbeta = sk->dbeta; g.sink.coef = sk->coef;
        }
    }
    // one launch per number of live depth taps (see k_wino: plane_list), most taps first
    LISEC_CHECK_ARG(g.Do <= 8, "Winograd form: at most 8 output planes");
    int live[8];
    for (int d = 0; d < g.Do; ++d) {
        live[d] = 0;
        for (int kd = 0; kd < g.KD; ++kd) {
            if (c->mode == 0) { const int sd = (d << g.ls_d) - g.pd + kd; live[d] += sd >= 0 && sd < g.Di; }
            else { const int t = d + g.pd - kd; live[d] += t >= 0 && (t & ((1 << g.ls_d) - 1)) == 0 && (t >> g.ls_d) < g.Di; }
        }
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const int exp = (flags >> 16) & 7;               // timing-only variants (tools/wino_stamps.py), 0 in every real call
#define LISEC_WINO_GO(X_, E_) LISEC_LAUNCH((k_wino<X_, E_>), grid, dim3(kWinoThreads), kWinoLds, st, g, c->mode, in, wino_w, bias, \
        in_bnstate, flags, out, BH, BW, plane_list)
    for (int taps = g.KD; taps >= 0; --taps) {
        unsigned plane_list = 0;
        int nslots = 0;
        for (int d = 0; d < g.Do; ++d)
            if (live[d] == taps) plane_list |= (unsigned)d << (4 * nslots++);
        if (!nslots) continue;
        dim3 grid(nslots * BH * BW, nnb, 1);
        if (exp == 1) LISEC_WINO_GO(false, 1);
        else if (exp == 2) LISEC_WINO_GO(false, 2);
        else if (exp == 3) LISEC_WINO_GO(false, 3);
        else if (exp == 4) LISEC_WINO_GO(false, 4);
        else if (exp == 5) LISEC_WINO_GO(false, 5);
        else if (exp == 6) LISEC_WINO_GO(false, 6);
        else if (flags & LISEC_CONV_TAG_ROOFLINE)
            LISEC_LAUNCH((k_wino<false, 0, 1>), grid, dim3(kWinoThreads), kWinoLds, st, g, c->mode, in, wino_w, bias, in_bnstate, flags, out, BH, BW, plane_list);
        else if (in_bnstate) LISEC_WINO_GO(true, 0);
        else LISEC_WINO_GO(false, 0);
    }
#undef LISEC_WINO_GO
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

// Diagnostic: points k_wino's stamp buffer at `buf` (device, 8192*8 uint64) or NULL.
extern "C" int lisec_debug_wino_stamps(unsigned long long* buf) {
    LISEC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_wino_stamps), &buf, sizeof(buf)));
    return LISEC_OK;
}
