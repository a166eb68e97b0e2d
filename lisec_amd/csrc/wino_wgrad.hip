// Winograd F(2x2, 3x3) form of the WEIGHT gradient of the stride-1 3x3 (h, w) contractions (model_training.py:193, :299: what
// fit() derives for a Conv3D kernel) on the fp32 matrix cores of gfx950.
//
// Forward (wino.hip):  Y_t = A^T [ sum_c U[pt][c][n] . V_t[pt][c] ] A  per 2x2-output tile t, V_t = B^T d_t B, U = G g G^T.
// Hence   dL/dU[kd][pt][c][n] = sum_{do} sum_t V_t[pt][c] * dM_t[pt][n],   dM_t = A dY_t A^T   (4x4 from the tile's 2x2 dY),
// and     dL/dg[kd][kh][kw][c][n] = sum_{i,j} G[i][kh] dL/dU[kd][4 i + j][c][n] G[j][kw]     (G^T dU G)
// -- 16 contractions of (Cin x tiles) by (tiles x Cout) per depth tap: 4 / 9 of the MFMAs of the direct weight gradient
// (wgrad.hip), fp32 in / fp32 accumulate; B^T, A hold 0 and +-1, G holds 1 and +-1/2.
//
// k_wino_wgrad: 255-256 resident 512-thread workgroups (one per CU; wave roles as in k_wino: two waves per SIMD, waves 0-3
// stage).  A workgroup owns ONE depth tap kd and accumulates the whole dU[kd] (16 points x 64 x 64 = 128 accumulator registers
// per lane) over its share of the (output plane, 8 x 8-tile block) units of that tap; the reduction dimension is the TILE: a
// chunk is 8 tiles (2 x 4), per chunk the 6 x 10 x 64-channel x window and the 4 x 8 x 64-channel dY window sit in LDS
// (fetched through buffer descriptors: rows outside the map are out-of-range zeros), every staging thread transforms one
// (tile pair, channel) of each operand and stores 16 x 8 bytes of V[pt][c][8 tiles] / dM[pt][n][8 tiles], then 16 points x 4 K
// steps of v_mfma_f32_32x32x2_f32 per (c, n) quadrant -- the MFMA loop, the LDS layout and its swizzle are k_wino's.  At the
// end every workgroup stores its accumulators as a slab in the register layout; k_wino_wgrad_sum adds the slabs of a tap in a
// fixed order (deterministic) and k_wino_wgrad_finish applies G^T . G and writes the Keras kernel layout (taps, Cin, Cout).
#include <type_traits>

#include "conv.h"

namespace lisec {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreadsW = 512;
constexpr int WK = 8;                        // tiles per K chunk (2 rows x 4 columns of tiles)
constexpr int OP_FLOATS = 16 * 64 * WK;      // one operand image: [pt][64 rows][8]
constexpr int STAGE_FLOATS = 2 * OP_FLOATS;  // V image + dM image
constexpr int XW_ROWS = 6, XW_COLS = 10, DW_ROWS = 4, DW_COLS = 8;
constexpr int W_STRIDE = 68;                 // floats per window position: 64 channels + 4 pad
constexpr int XW_FLOATS = XW_ROWS * XW_COLS * W_STRIDE, DW_FLOATS = DW_ROWS * DW_COLS * W_STRIDE;
constexpr size_t kWgradLds = (2 * STAGE_FLOATS + XW_FLOATS + DW_FLOATS) * sizeof(float);      // 156 096
constexpr int SLAB_FLOATS = 16 * 64 * 64;    // dU of one tap: 8 waves x 8 points x 16 registers x 64 lanes

__device__ __forceinline__ int swz(int row, int k) {
    return row * 8 + ((((k >> 2) ^ ((row >> 3) & 1))) << 2) + (k & 3);
}

struct WgradArgs {
    const float* x; const float* dy; float* slabs;
    int Di, Do, H, W, KD, ls_d, pd;
    int BH, BW;                      // blocks of 8 x 8 tiles per plane
    int nkd;                         // live depth taps
    unsigned kd_list;                // their indices, one nibble each
    int nwg;
};

// units of tap kd: (output plane do whose tap kd reads a plane inside x) x (block); returns the count and the plane mask
__host__ __device__ inline int tap_planes(int kd, int Do, int Di, int ls_d, int pd, unsigned* mask) {
    unsigned m = 0;
    int n = 0;
    for (int d = 0; d < Do; ++d) {
        const int s = (d << ls_d) - pd + kd;
        if (s >= 0 && s < Di) { m |= 1u << d; ++n; }
    }
    *mask = m;
    return n;
}

__global__ void __launch_bounds__(kThreadsW) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_wino_wgrad(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves 0-3 (half 0) stage the x operand (window + B^T d B), waves 4-7 (half 1) the dY operand (window + A dY A^T); wave w
    // and wave w + 4 sit on one SIMD and own the same 32 x 32 (c, n) quadrant, points 0-7 and 8-15
    const int half = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int tid = threadIdx.x & 255;               // staging identity inside the half
    const int H = a.H, W = a.W;
    float* sXW = smem + 2 * STAGE_FLOATS;
    float* sDW = sXW + XW_FLOATS;
    // ---- this workgroup's tap and its share of the units ---------------------------------------------------------------
    const int kslot = blockIdx.x % a.nkd, wk = blockIdx.x / a.nkd;
    const int nwk = (a.nwg - kslot + a.nkd - 1) / a.nkd;         // workgroups of this tap
    const int kd = (a.kd_list >> (4 * kslot)) & 15;
    unsigned do_mask;
    const int ndo = tap_planes(kd, a.Do, a.Di, a.ls_d, a.pd, &do_mask);
    const int per_plane = a.BH * a.BW;
    const int units = ndo * per_plane;
    const int nunits = wk < units ? (units - wk + nwk - 1) / nwk : 0;
    const int nchunks = nunits * 8;
    const size_t plane = (size_t)H * W * 64;
    // ---- loader identities: x window 60 positions x 16 quads = 960 sixteen-byte items (4 per thread of half 0), dY window
    // 32 x 16 = 512 (2 per thread of half 1).  Byte offset of an item inside ONE plane relative to the window's first position;
    // rows outside the map fall outside the plane's descriptor (zeros), columns outside it are turned away explicitly (the low
    // 4 bits of an x item's offset carry its window column)
    int xw_rel[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int item = tid + 256 * r;
        item = item < XW_ROWS * XW_COLS * 16 ? item : XW_ROWS * XW_COLS * 16 - 1;      // (surplus threads repeat the last item)
        const int pos = item >> 4, quad = item & 15;
        const int row = pos / XW_COLS, col = pos - row * XW_COLS;
        xw_rel[r] = (((row * W + col) * 64 + quad * 4) * 4) | col;
    }
    const int w_lds0 = (tid >> 4) * W_STRIDE + (tid & 15) * 4;                         // item tid + 256 r: + 16 W_STRIDE r
    const int xw_lds3 = tid + 768 < XW_ROWS * XW_COLS * 16 ? w_lds0 + 48 * W_STRIDE : (XW_ROWS * XW_COLS - 1) * W_STRIDE + 60;
    const int dcol = (tid >> 4) & 7, drow = tid >> 7;                                  // dY item tid + 256 r: row drow + 2 r
    const int dw_rel0 = ((drow * W + dcol) * 64 + (tid & 15) * 4) * 4;                 // second item: + 2 W rows
    // ---- transform identity: (tile pair, channel) ----------------------------------------------------------------------------
    const int ch = tid & 63, tp = tid >> 6;          // tile pair: tile row tp >> 1, tile columns 2 (tp & 1), 2 (tp & 1) + 1
    const int tyl = tp >> 1, tq = tp & 1;
    const int xsrc = ((2 * tyl) * XW_COLS + 4 * tq) * W_STRIDE + ch;       // patch (0, 0) of the pair's first tile
    const int dsrc = ((2 * tyl) * DW_COLS + 4 * tq) * W_STRIDE + ch;
    const int odst = ch * 8 + ((tyl ^ ((ch >> 3) & 1)) << 2) + 2 * tq;      // + pt * 512: rows = channel, k = tile 4 tyl + 2 tq (+1)

    // ---- chunk walk: flat chunk index -> (unit, sub-block) -> plane base and the window origin ---------------------------
    // (the unit is decoded -- two scalar divisions -- only when the walk enters a new one, every eighth chunk)
    const float* u_base = a.x;
    int u_by = 0, u_bx = 0;
    auto enter_unit = [&](int unit, int role) {
        const int u = wk + unit * nwk;
        const int di = u / per_plane, blk = u - di * per_plane;
        u_by = blk / a.BW; u_bx = blk - u_by * a.BW;
        int d = 0, seen = 0;                         // the di-th set bit of do_mask
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool on = (do_mask >> b) & 1;
            if (on && seen == di) d = b;
            seen += on ? 1 : 0;
        }
        const int sd = (d << a.ls_d) - a.pd + kd;
        u_base = role == 0 ? a.x + (size_t)sd * plane : a.dy + (size_t)d * plane;
    };
    float4 wr0, wr1, wr2, wr3;                       // this half's next window on its way to LDS (half 1 uses two)
    auto load_window = [&](int cidx, auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        cidx = cidx < nchunks ? cidx : nchunks - 1;  // (past the end the walk stays on the last chunk: its image is never read)
        if ((cidx & 7) == 0) enter_unit(cidx >> 3, ROLE);
        const int sub = cidx & 7;
        const int y0 = 16 * u_by + 4 * (sub >> 1), x0 = 16 * u_bx + 8 * (sub & 1);     // first OUTPUT row / column of the 2 x 4 tiles
        const int soff = (y0 * W + x0) * 64 * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(u_base), 0, (int)(plane * 4), 0x00020000);
#define WG_LD(R_, OFF_, OK_) { const u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rs, (OK_) ? (OFF_) : -1, 0, 0);       \
                               R_ = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); }
        if (ROLE == 0) {
            // window origin = (y0 - 1, x0 - 1): offsets below zero wrap past num_records (zeros), offsets beyond the plane too
            const int xo = soff - (W + 1) * 64 * 4;
            WG_LD(wr0, xo + (xw_rel[0] & ~15), (unsigned)(x0 - 1 + (xw_rel[0] & 15)) < (unsigned)W)
            WG_LD(wr1, xo + (xw_rel[1] & ~15), (unsigned)(x0 - 1 + (xw_rel[1] & 15)) < (unsigned)W)
            WG_LD(wr2, xo + (xw_rel[2] & ~15), (unsigned)(x0 - 1 + (xw_rel[2] & 15)) < (unsigned)W)
            WG_LD(wr3, xo + (xw_rel[3] & ~15), (unsigned)(x0 - 1 + (xw_rel[3] & 15)) < (unsigned)W)
        } else {
            WG_LD(wr0, soff + dw_rel0, x0 + dcol < W && y0 + drow < H)
            WG_LD(wr1, soff + dw_rel0 + 2 * W * 64 * 4, x0 + dcol < W && y0 + drow + 2 < H)
        }
#undef WG_LD
    };
    auto store_window = [&](auto role_tag) {
        if (decltype(role_tag)::value == 0) {
            *reinterpret_cast<float4*>(sXW + w_lds0) = wr0; *reinterpret_cast<float4*>(sXW + w_lds0 + 16 * W_STRIDE) = wr1;
            *reinterpret_cast<float4*>(sXW + w_lds0 + 32 * W_STRIDE) = wr2; *reinterpret_cast<float4*>(sXW + xw_lds3) = wr3;
        } else {
            *reinterpret_cast<float4*>(sDW + w_lds0) = wr0; *reinterpret_cast<float4*>(sDW + w_lds0 + 16 * W_STRIDE) = wr1;
        }
    };
    // ---- the transform of one (tile pair, channel): 24 (x) or 8 (dY) window reads, sixteen 8-byte stores --------------------------
    float px[24];
    auto read_x = [&](int i0, int n) {               // patch rows i0 .. i0 + n - 1 (6 columns each)
#pragma unroll
        for (int i = i0; i < i0 + n; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) px[6 * i + j] = sXW[xsrc + (i * XW_COLS + j) * W_STRIDE];
    };
    auto read_d = [&]() {                            // (the dY half keeps its 2 x 4 patch in px[0..7])
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) px[4 * i + j] = sDW[dsrc + (i * DW_COLS + j) * W_STRIDE];
    };
    float vA[16], vB[16];                            // the transform of the pair's two tiles
    // (macros, not lambdas taking a pointer: an array whose address is passed around stays in scratch memory)
#define WG_BT_D_B(J0_, V_)  /* B^T d B of the tile whose patch starts at column J0_ of px */                                   \
    {                                                                                                                        \
        float t_[16];                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
            const float d0 = px[0 + (J0_) + j], d1 = px[6 + (J0_) + j], d2 = px[12 + (J0_) + j], d3 = px[18 + (J0_) + j];    \
            t_[0 + j] = d0 - d2; t_[4 + j] = d1 + d2; t_[8 + j] = d2 - d1; t_[12 + j] = d1 - d3;                             \
        }                                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                      \
            V_[4 * i + 0] = t_[4 * i + 0] - t_[4 * i + 2];                                                                   \
            V_[4 * i + 1] = t_[4 * i + 1] + t_[4 * i + 2];                                                                   \
            V_[4 * i + 2] = t_[4 * i + 2] - t_[4 * i + 1];                                                                   \
            V_[4 * i + 3] = t_[4 * i + 1] - t_[4 * i + 3];                                                                   \
        }                                                                                                                    \
    }
#define WG_A_DY_AT(J0_, V_)  /* A dY A^T of the tile whose 2 x 2 gradient starts at column J0_ of px */                        \
    {                                                                                                                        \
        const float g00 = px[(J0_)], g01 = px[(J0_) + 1], g10 = px[4 + (J0_)], g11 = px[4 + (J0_) + 1];                      \
        float h_[8];                                                                                                         \
        h_[0] = g00; h_[1] = g01; h_[2] = g00 + g10; h_[3] = g01 + g11; h_[4] = g00 - g10; h_[5] = g01 - g11;                \
        h_[6] = -g10; h_[7] = -g11;                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                      \
            V_[4 * i + 0] = h_[2 * i];                                                                                       \
            V_[4 * i + 1] = h_[2 * i] + h_[2 * i + 1];                                                                       \
            V_[4 * i + 2] = h_[2 * i] - h_[2 * i + 1];                                                                       \
            V_[4 * i + 3] = -h_[2 * i + 1];                                                                                  \
        }                                                                                                                    \
    }
    auto store_pairs = [&](float* img, int p0, int n) {          // points p0 .. p0 + n - 1 of both tiles of the pair
#pragma unroll
        for (int p = p0; p < p0 + n; ++p)
            *reinterpret_cast<float2*>(img + p * 512 + odst) = make_float2(vA[p], vB[p]);
    };
    auto transform_all = [&](float* stage, auto role_tag) {         // (prologue: the whole side work of one chunk at once)
        if (decltype(role_tag)::value == 0) { WG_BT_D_B(0, vA) WG_BT_D_B(2, vB) store_pairs(stage, 0, 16); }
        else { WG_A_DY_AT(0, vA) WG_A_DY_AT(2, vB) store_pairs(stage + OP_FLOATS, 0, 16); }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = (f32x16){0};
    const int arow = wm * 32 + (lane & 31), brow = wn * 32 + (lane & 31);
    const int aoff = half * 8 * 512 + arow * 8 + (((lane >> 5) ^ ((arow >> 3) & 1)) << 2);
    const int boff = OP_FLOATS + half * 8 * 512 + brow * 8 + (((lane >> 5) ^ ((brow >> 3) & 1)) << 2);

    auto prologue = [&](auto role_tag) {
        constexpr int ROLE = decltype(role_tag)::value;
        load_window(0, role_tag);
        store_window(role_tag);
        __syncthreads();
        if (ROLE == 0) read_x(0, 4); else read_d();
        load_window(1, role_tag);
        __syncthreads();                             // every patch of window 0 is in registers
        store_window(role_tag);
        transform_all(smem, role_tag);
    };
    if (nchunks > 0) {
        if (half == 0) prologue(std::integral_constant<int, 0>{}); else prologue(std::integral_constant<int, 1>{});
    }
    __syncthreads();
    // Chunk c: 32 MFMAs per wave on stage c & 1; meanwhile each half builds ITS operand's image of chunk c + 1 from its window
    // in LDS and fetches its window of chunk c + 2 -- in 8 groups of one point (two fragment reads, four MFMAs) carrying two
    // slices p of the side work each (an fp32 MFMA shares its SIMD's issue with nothing: wino.hip), pinned by sched_barrier:
    //   p 0      global loads of the window of chunk c + 2 (four / two 16-byte loads through the plane's descriptor)
    //   p 0-3    the patch of chunk c + 1 from the window (x: 4 rows x 6 positions; dY: 2 x 4 at p 3)
    //   p 4-5    the transform of the pair's first tile, of its second;  p 6-7  their 16 eight-byte stores
    //   -- barrier: every wave has read the windows --
    //   p 11     the window of chunk c + 2 -> LDS
    auto chunk = [&](int c, auto par_tag, auto role_tag) {
        constexpr int PAR = decltype(par_tag)::value, ROLE = decltype(role_tag)::value;
        const float* st = smem + PAR * STAGE_FLOATS;
        float* nx = smem + (PAR ^ 1) * STAGE_FLOATS + (ROLE == 0 ? 0 : OP_FLOATS);
        const float* ap = st + aoff;
        const float* bp = st + boff;
        float4 a0 = *reinterpret_cast<const float4*>(ap);
        float4 b0 = *reinterpret_cast<const float4*>(bp);
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            float4 a0n = a0, b0n = b0;
            if (pp + 1 < 8) {
                a0n = *reinterpret_cast<const float4*>(ap + (pp + 1) * 512);
                b0n = *reinterpret_cast<const float4*>(bp + (pp + 1) * 512);
            }
#pragma unroll
            for (int p = 2 * pp; p < 2 * pp + 2; ++p) {
                if (p == 0) load_window(c + 2, role_tag);
                if (ROLE == 0) {
                    if (p < 4) read_x(p, 1);
                    if (p == 4) WG_BT_D_B(0, vA)
                    if (p == 5) WG_BT_D_B(2, vB)
                } else {
                    if (p == 3) read_d();
                    if (p == 4) WG_A_DY_AT(0, vA)
                    if (p == 5) WG_A_DY_AT(2, vB)
                }
                if (p == 6) store_pairs(nx, 0, 8);
                if (p == 7) store_pairs(nx, 8, 8);
                if (p == 11) store_window(role_tag);
            }
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[pp], 0, 0, 0);
            acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[pp], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (pp == 3) __syncthreads();            // every wave has read the windows of chunk c + 1
            a0 = a0n; b0 = b0n;
        }
        __syncthreads();
    };
    if (half == 0) {
        for (int c = 0; c < nchunks; c += 2) {
            chunk(c, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            chunk(c + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        }
    } else {
        for (int c = 0; c < nchunks; c += 2) {
            chunk(c, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            chunk(c + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
    }
    // ---- the slab: accumulators in the register layout, [workgroup][wave][point][register][lane] ---------------------------------------
    float* slab = a.slabs + (size_t)blockIdx.x * SLAB_FLOATS + (size_t)wave * (8 * 16 * 64) + lane;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(p * 16 + r) * 64] = acc[p][r];
}

// sum[kslot][e] = sum over the workgroups of tap kslot of their slab element e, in workgroup order (deterministic)
__global__ void k_wino_wgrad_sum(const float* __restrict__ slabs, int nwg, int nkd, float* __restrict__ sum) {
    const int e4 = blockIdx.x * 256 + threadIdx.x;               // float4 index inside a slab
    const int kslot = blockIdx.y;
    if (e4 >= SLAB_FLOATS / 4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = kslot; w < nwg; w += nkd) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)w * SLAB_FLOATS + e4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(sum + (size_t)kslot * SLAB_FLOATS + e4 * 4) = s;
}

// dW[kd][kh][kw][c][n] = G^T dU[kd] G from the summed slabs (taps without a live plane: zeros); dst element (tap, c, n) at
// dst[tap * 4096 * .. ] in the Keras layout (taps, 64, 64)
__global__ void k_wino_wgrad_finish(const float* __restrict__ sum, int KD, int nkd, unsigned kd_list, float* __restrict__ dW) {
    const int idx = blockIdx.x * 256 + threadIdx.x;              // (kd, c, n)
    if (idx >= KD * 4096) return;
    const int kd = idx >> 12, c = (idx >> 6) & 63, n = idx & 63;
    int kslot = -1;
    for (int s = 0; s < nkd; ++s)
        if ((int)((kd_list >> (4 * s)) & 15) == kd) kslot = s;
    float u[16];
#pragma unroll
    for (int pt = 0; pt < 16; ++pt) {
        const int wave = (pt >> 3) * 4 + (c >> 5) * 2 + (n >> 5), p = pt & 7;
        const int cl = c & 31, hi = (cl >> 2) & 1, r = (cl & 3) + 4 * (cl >> 3);
        const int lane = hi * 32 + (n & 31);
        u[pt] = kslot < 0 ? 0.f : sum[(size_t)kslot * SLAB_FLOATS + ((wave * 8 + p) * 16 + r) * 64 + lane];
    }
    // G^T u G, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = u[0 + j] + 0.5f * (u[4 + j] + u[8 + j]);
        t[1][j] = 0.5f * (u[4 + j] - u[8 + j]);
        t[2][j] = 0.5f * (u[4 + j] + u[8 + j]) + u[12 + j];
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const float w0 = t[kh][0] + 0.5f * (t[kh][1] + t[kh][2]);
        const float w1 = 0.5f * (t[kh][1] - t[kh][2]);
        const float w2 = 0.5f * (t[kh][1] + t[kh][2]) + t[kh][3];
        float* d = dW + ((size_t)((kd * 3 + kh) * 3) * 64 + c) * 64 + n;
        d[0] = w0; d[4096] = w1; d[2 * 4096] = w2;
    }
}

bool wgrad_wino_ok(const lisec_conv_geom* c, const char** msg) {
#define NEED(cond, text) do { if (!(cond)) { *msg = text; return false; } } while (0)
    NEED(c->mode == 0, "mode 0 (the geometry of the forward layer)");
    NEED(c->KH == 3 && c->KW == 3 && c->sh == 1 && c->sw == 1 && c->ph == 1 && c->pw == 1, "3x3 (h, w) taps, stride 1, pad 1");
    NEED(c->Hi == c->Ho && c->Wi == c->Wo && c->Ho >= 2 && c->Wo >= 2, "equal input and output maps");
    NEED(c->Cin == 64 && c->Cout == 64 && c->in_stride == 64 && c->out_stride == 64, "64 -> 64 channels, packed rows");
    NEED(c->KD <= 4 && c->Do <= 8 && !c->ps, "at most 4 depth taps and 8 output planes");
    NEED((long long)c->Hi * c->Wi * 64 < (1LL << 29), "one plane below 2 GB");
#undef NEED
    return true;
}

int live_taps(const ConvGeom& g, unsigned* kd_list) {
    int n = 0;
    unsigned list = 0, m;
    for (int kd = 0; kd < g.KD; ++kd)
        if (tap_planes(kd, g.Do, g.Di, g.ls_d, g.pd, &m) > 0) list |= (unsigned)kd << (4 * n++);
    *kd_list = list;
    return n;
}

int wgrad_nwg(int nkd) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return nkd > 0 ? cus / nkd * nkd : cus;          // the same number of workgroups for every tap
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" int lisec_conv_wgrad_winograd_supported(const lisec_conv_geom* c) {
    ConvGeom g;
    if (!c || conv_geom_check(c, &g)) return 0;
    const char* msg = "";
    return wgrad_wino_ok(c, &msg) ? 1 : 0;
}

extern "C" size_t lisec_conv_wgrad_winograd_workspace_bytes(const lisec_conv_geom* c) {
    ConvGeom g;
    if (!c || conv_geom_check(c, &g)) return 0;
    unsigned list;
    const int nkd = live_taps(g, &list);
    return ((size_t)wgrad_nwg(nkd) + (size_t)(nkd > 0 ? nkd : 1)) * SLAB_FLOATS * sizeof(float);
}

extern "C" int lisec_conv_wgrad_winograd(const lisec_conv_geom* c, const float* in, const float* dy, void* workspace,
                                         size_t workspace_bytes, float* dW, lisec_stream_t stream_) {
    ConvGeom g;
    if (int rc = conv_geom_check(c, &g)) return rc;
    const char* msg = "";
    LISEC_CHECK_ARG(wgrad_wino_ok(c, &msg), "Winograd weight gradient needs: %s", msg);
    LISEC_CHECK_ARG(in && dy && dW && workspace, "NULL pointer");
    LISEC_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dW & 3) == 0,
                    "operands and workspace: 16-byte aligned");
    LISEC_CHECK_ARG(workspace_bytes >= lisec_conv_wgrad_winograd_workspace_bytes(c), "workspace too small");
    WgradArgs a;
    a.x = in; a.dy = dy; a.slabs = static_cast<float*>(workspace);
    a.Di = g.Di; a.Do = g.Do; a.H = g.Ho; a.W = g.Wo; a.KD = g.KD; a.ls_d = g.ls_d; a.pd = g.pd;
    const int TH = (g.Ho + 1) / 2, TW = (g.Wo + 1) / 2;
    a.BH = cdiv(TH, 8); a.BW = cdiv(TW, 8);
    a.nkd = live_taps(g, &a.kd_list);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    float* sum = a.slabs;
    if (a.nkd > 0) {
        a.nwg = wgrad_nwg(a.nkd);
        sum = a.slabs + (size_t)a.nwg * SLAB_FLOATS;
        LISEC_LAUNCH(k_wino_wgrad, dim3(a.nwg), dim3(kThreadsW), kWgradLds, st, a);
        LISEC_LAUNCH(k_wino_wgrad_sum, dim3(SLAB_FLOATS / 4 / 256, a.nkd), dim3(256), 0, st, (const float*)a.slabs, a.nwg, a.nkd, sum);
    }
    LISEC_LAUNCH(k_wino_wgrad_finish, dim3(cdiv(g.KD * 4096, 256)), dim3(256), 0, st, (const float*)sum, g.KD, a.nkd, a.kd_list, dW);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
