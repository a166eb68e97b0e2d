// Weight gradients of every dense contraction on the fp32 matrix cores (gfx950).
//
//   dW[tap][c][n] = sum_m  A_tap[m][c] * dY[m][n]        (what Keras' fit() derives by autograd for
//   Conv3D / Conv2D / Conv2DTranspose / Dense kernels, model_training.py:184,193,203,246-255,:299)
// A_tap is the same on-the-fly gather (+ BatchNormalization/ReLU of the producing layer applied on
// load) as in igemm.hip; the contraction runs over output positions m, so the GEMM is
// (64 c) x (64 n) with K = M: tiny output, huge K.  Work split:
//   workgroup = (group of KW consecutive taps) x (64-channel c block) x (64-channel n block)
//               x (one of S ranges of 128-row M tiles)
// The dY tile is staged once per M tile and reused by the KW taps of the group; each wave keeps one
// 32x32 accumulator per tap for its whole M range and writes it once to a partial slab
// [S][tap][Cin][Cout]; a second kernel sums the S slabs in index order (deterministic, no atomics)
// into the Keras kernel layout.
#include "conv.h"


namespace lisec {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BMW = 128, BC = 64;                // rows per M tile, channel block
constexpr int kThreads = 256;
constexpr int TILE_FLOATS = BMW * BC;            // 8192 (32 KB)

// diagnostic (tools/wgrad_stamps.py): 100 MHz s_memrealtime stamps of thread 0 of every workgroup at its phase boundaries
// (0 entry, 1 first tile in LDS, 2 loop done, 3 slabs stored; 5 = tiles run, 6 = HW_ID | XCC_ID << 32; 4 / 7 = s_memtime
// shader cycles at 1 / 2: the in-kernel clock of the loop); nullptr = off
__device__ unsigned long long* g_wgrad_stamps = nullptr;
#define WGRAD_STAMP(K_)                                                                                    \
    do {                                                                                                   \
        if (stamps && threadIdx.x == 0 && stamp_wg < 8192) {                                               \
            stamps[(size_t)stamp_wg * 8 + (K_)] = __builtin_amdgcn_s_memrealtime();                        \
            if ((K_) == 1) stamps[(size_t)stamp_wg * 8 + 4] = __builtin_amdgcn_s_memtime();    /* shader cycles: 4, 7 */  \
            if ((K_) == 2) stamps[(size_t)stamp_wg * 8 + 7] = __builtin_amdgcn_s_memtime();                \
            if ((K_) == 0)                                                                                 \
                stamps[(size_t)stamp_wg * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | \
                                                   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
        }                                                                                                  \
    } while (0)

// RL: row-list instantiation (tiles are tested for work by their decoded row masks, see first_live)
template <int MODE, int TG, bool RL = false>
__global__ void __launch_bounds__(kThreads)
k_wgrad(ConvGeom g, const float* __restrict__ in, const float* __restrict__ in_bn, int flags,
        const float* __restrict__ dy, const float* __restrict__ dy_bn, int nsplit, int tiles_per_split,
        float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned long long* stamps = g_wgrad_stamps;
    const unsigned stamp_wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    WGRAD_STAMP(0);
    float* sA = smem;
    float* sD = smem + TILE_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = g.KD * g.KH * (g.KW / TG);
    const int split = blockIdx.x / ngroups, group = blockIdx.x - split * ngroups;
    const int tap0 = group * TG;                 // taps tap0 .. tap0+TG-1 share (kd, kh)
    const int c0 = blockIdx.y * BC, n0 = blockIdx.z * BC;
    const int HW = g.Ho * g.Wo;
    const int mlimit = row_limit(g);
    const int ntiles = (mlimit + BMW - 1) / BMW;
    // dense rows: split s owns the tiles [s * tiles_per_split, ...).  Row lists: the launch is sized for the CAPACITY but
    // only the tiles below the device-side count hold rows, so split s takes tiles s, s + nsplit, s + 2 nsplit, ... --
    // every split gets its share of the live tiles whatever their number (84 000 of 200 000 rows: contiguous ranges left
    // 58 % of the workgroups without a row and 14 tiles to each of the others)
    const int ts = RL ? nsplit : 1;
    const int t_begin = RL ? split : split * tiles_per_split;
    const int t_end = RL ? ntiles : (t_begin + tiles_per_split < ntiles ? t_begin + tiles_per_split : ntiles);
    const int kd = tap0 / (g.KH * g.KW), kh = (tap0 / g.KW) % g.KH, kw0 = tap0 % g.KW;

    const int piece = tid & 15;
    const int cA = c0 + piece * 4, cD = n0 + piece * 4;
    const bool cokA = cA < g.Cin, cokD = cD < g.Cout;
    float4 tsc = make_float4(1, 1, 1, 1), tsh = make_float4(0, 0, 0, 0);
    if (in_bn && cokA) {
        tsc = *reinterpret_cast<const float4*>(in_bn + cA);
        tsh = *reinterpret_cast<const float4*>(in_bn + g.Cin + cA);
    }
    const float relu_lo = (flags & LISEC_CONV_IN_RELU) ? 0.f : -INFINITY;
    float4 dsc = make_float4(1, 1, 1, 1), dsh = make_float4(0, 0, 0, 0);
    if (dy_bn && cokD) {
        dsc = *reinterpret_cast<const float4*>(dy_bn + cD);
        dsh = *reinterpret_cast<const float4*>(dy_bn + g.Cout + cD);
    }
    const float drelu_lo = (flags & LISEC_CONV_DY_RELU) ? 0.f : -INFINITY;
    unsigned dvalid = 0;

    f32x16 acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) acc[t] = (f32x16){0};

    float4 ra[8], rd[8];
    unsigned valid_mask = 0;
    RowGather rows[8];

    auto decode_rows = [&](int tile) {
#pragma unroll
        for (int p = 0; p < 8; ++p) rows[p] = row_gather(g, tile * BMW + p * 16 + (tid >> 4), MODE, cA);
    };
    auto tile_live = [&](int tile) -> bool {     // whole tile outside the valid depth range of this tap group
        if (g.row_coords) return true;
        int mfirst = tile * BMW, mlast = mfirst + BMW - 1 < g.M ? mfirst + BMW - 1 : g.M - 1;
        int df = mfirst / HW, dl = mlast / HW;
        if (df != dl) return true;
        int tmp;
        return (axis_mask(df, g.KD, g.ls_d, g.pd, g.Di, MODE, tmp) >> kd) & 1;
    };
    auto load_a = [&](int kw) {
        const int soff = tap_delta(g, kd, kh, kw, MODE);
        const int tbits = cokA ? tap_bits(kd, kh, kw) : 0x7fffffff;
        valid_mask = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const bool ok = (rows[p].mask & tbits) == tbits;
            const int off = ok ? rows[p].off + soff : 0;
            ra[p] = *reinterpret_cast<const float4*>(in + off);
            valid_mask |= ok ? (1u << p) : 0u;
        }
    };
    auto load_d = [&](int tile) {
        dvalid = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int m = tile * BMW + p * 16 + (tid >> 4);
            const bool ok = m < mlimit && cokD;                      // branch-free: rows beyond the layer read element 0
            rd[p] = *reinterpret_cast<const float4*>(dy + (ok ? (size_t)m * g.out_stride + cD : (size_t)0));
            dvalid |= ok ? (1u << p) : 0u;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float4 v = ra[p];
            const bool ok = (valid_mask >> p) & 1;
            v.x = ok ? fmaxf(fmaf(v.x, tsc.x, tsh.x), relu_lo) : 0.f;
            v.y = ok ? fmaxf(fmaf(v.y, tsc.y, tsh.y), relu_lo) : 0.f;
            v.z = ok ? fmaxf(fmaf(v.z, tsc.z, tsh.z), relu_lo) : 0.f;
            v.w = ok ? fmaxf(fmaf(v.w, tsc.w, tsh.w), relu_lo) : 0.f;
            *reinterpret_cast<float4*>(sA + (p * 16 + (tid >> 4)) * BC + piece * 4) = v;
        }
    };
    auto store_d = [&]() {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float4 v = rd[p];
            const bool ok = (dvalid >> p) & 1;
            v.x = ok ? fmaxf(fmaf(v.x, dsc.x, dsh.x), drelu_lo) : 0.f;
            v.y = ok ? fmaxf(fmaf(v.y, dsc.y, dsh.y), drelu_lo) : 0.f;
            v.z = ok ? fmaxf(fmaf(v.z, dsc.z, dsh.z), drelu_lo) : 0.f;
            v.w = ok ? fmaxf(fmaf(v.w, dsc.w, dsh.w), drelu_lo) : 0.f;
            *reinterpret_cast<float4*>(sD + (p * 16 + (tid >> 4)) * BC + piece * 4) = v;
        }
    };

    // MFMA 32x32x2: A operand lane(i = c, h) <- A_tap[m = 8*kk + 4h + j][c]; B operand lane(h, n) <- dY[m][n]
    const float* aCol = sA + (4 * (lane >> 5)) * BC + (wave >> 1) * 32 + (lane & 31);
    const float* dCol = sD + (4 * (lane >> 5)) * BC + (wave & 1) * 32 + (lane & 31);

    // step list: (tile, t) with t in [0, TG); dY is staged with t == 0
    // first tile >= `tile` with work for this tap group; its row descriptors are decoded on return.  Row lists: the
    // voxels are sorted by cell, so most tiles share one depth (parity) and whole (kd, kh) groups fall away
    const int gbits = (1 << kd) | (16 << kh);
    auto first_live = [&](int tile) -> int {
        if (!RL) {
            while (tile < t_end && !tile_live(tile)) ++tile;
            if (tile < t_end) decode_rows(tile);
            return tile;
        }
        for (; tile < t_end; tile += ts) {
            decode_rows(tile);
            int pred = 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) pred |= (rows[p].mask & gbits) == gbits;
            if (__syncthreads_or(pred)) break;
        }
        return tile;
    };
    int tile = first_live(t_begin);
    int t = 0;
    if (tile < t_end) {
        load_a(kw0);
        load_d(tile);
        store_a();
        store_d();
    }
    __syncthreads();
    WGRAD_STAMP(1);
    int stamp_tiles = 0;
    while (tile < t_end) {
        stamp_tiles += t == 0;
        // next step
        int ntile = tile, nt = t + 1;
        if (nt == TG) {
            nt = 0;
            ntile = first_live(ntile + ts);
        }
        const bool more = ntile < t_end;
        if (more) {
            if (nt == 0) load_d(ntile);
            load_a(kw0 + nt);
        }
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            if (tt == t) {
#pragma unroll 4
                for (int kk = 0; kk < BMW / 8; ++kk) {
                    const float* ap = aCol + kk * 8 * BC;
                    const float* dp = dCol + kk * 8 * BC;
                    const float a0 = ap[0], a1 = ap[BC], a2 = ap[2 * BC], a3 = ap[3 * BC];
                    const float b0 = dp[0], b1 = dp[BC], b2 = dp[2 * BC], b3 = dp[3 * BC];
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc[tt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) {
            store_a();
            if (nt == 0) store_d();
        }
        __syncthreads();
        tile = ntile;
        t = nt;
    }

    WGRAD_STAMP(2);
    // partial slab [split][tap][Cin][Cout]; C layout: col = lane&31 (n), row = (r&3)+8*(r>>2)+4*(lane>>5) (c)
    const int ntaps = g.KD * g.KH * g.KW;
#pragma unroll
    for (int tt = 0; tt < TG; ++tt) {
        float* base = partial + ((size_t)split * ntaps + tap0 + tt) * g.Cin * g.Cout;
        const int n = n0 + (wave & 1) * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (c < g.Cin && n < g.Cout) base[(size_t)c * g.Cout + n] = acc[tt][r];
        }
    }
    WGRAD_STAMP(3);
    if (stamps && threadIdx.x == 0 && stamp_wg < 8192) stamps[(size_t)stamp_wg * 8 + 5] = (unsigned long long)stamp_tiles;
}


// ---- halo variant: stride-1 (along w) 3-tap convolutions, dense positions -------------------------------------
// A tile is a run of up to 128 consecutive w' inside ONE output line (d',h'); for the workgroup's (kd,kh) the
// three kw taps read the same source line shifted by one position, so the source rows w0-pw .. w0-pw+len+1 are
// staged ONCE (zero outside the map) and tap kw reads LDS rows j+kw: half the global->LDS traffic of the generic
// kernel, no per-row address arithmetic, and 3x longer MFMA runs between barriers.  The contraction runs over the
// rows; the tiles of a line are made equal (W' = 400 -> 4 x 100 rows) so that no short tail tile pays a full staging round.

// The (kd, kh) groups a launch runs: groups no output line can read (depth stride 2 over a two-plane input: a third of
// them) get no workgroups at all.
struct LiveGroups { int n; unsigned char id[16]; };

// One weight-gradient contraction as the halo kernel sees it: single launches pass one, batched launches a table of them.
struct WgradItem {
    ConvGeom g;
    const float* in; const float* in_bn; const float* dy;
    float* partial;          // slabs (combine: in the accumulator register layout, behind the arrival counters)
    float* dW;               // combine: the finished kernel gradient is written by the last slice to arrive
    int* counters;           // combine: one arrival counter per (group, c block, n block), zero between calls
    int flags, nsplit, tiles_per_split, flip, LT, transpose, combine;
    int gx, gy, gz;          // grid of this item: gx = nsplit * groups
    LiveGroups live;
};

// Slab of one (slice, cell) in the register layout: [wave 4][q 12 = 3 taps x 4 float4][lane 64] float4 = 48 KB.
constexpr int kWSlabF4 = 4 * 12 * 64;
constexpr int kWgradCounters = 4096;

// NP: staging passes of 16 rows.  7 (tiles of <= 110 rows: the 400-wide middle layers run 4 x 100) needs 12 registers
// fewer than 9 and, with the fragment reads software-pipelined by hand, fits three workgroups per CU.
template <bool XF, int NP>
__device__ __forceinline__ void
wgrad_halo_body(const WgradItem& it, const int bx, const int by, const int bz, float* smem, unsigned long long* stamps,
                const unsigned stamp_wg) {
    constexpr int PA = NP, PD = NP == 7 ? 7 : 8;
    const ConvGeom& g = it.g;
    const float* __restrict__ in = it.in;
    const float* __restrict__ in_bn = it.in_bn;
    const float* __restrict__ dy = it.dy;
    float* __restrict__ partial = it.partial;
    const int flags = it.flags, nsplit = it.nsplit, tiles_per_split = it.tiles_per_split, flip = it.flip, LT = it.LT;
    const LiveGroups& live = it.live;
    (void)nsplit;
    // LT <= 128: rows per tile, chosen by the host so that the tiles of one line are equal (W' = 400 -> 4 x 100) and,
    // with DR = LT rounded up to 8, the workgroup's LDS is (2 DR + 2) x 256 B: 53 760 B at LT = 100, three per CU
    WGRAD_STAMP(0);
    const int DR = (LT + 7) & ~7;
    float* sA = smem;                                  // [DR + 2][64]: halo rows, zero beyond len + 2
    float* sD = smem + (DR + 2) * BC;                  // [DR][64]: dY rows, zero beyond len
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = live.n;
    const int split = bx / ngroups, gslot = bx - split * ngroups, group = live.id[gslot];
    const int kd = group / g.KH, kh = group - kd * g.KH;
    const int c0 = by * BC, n0 = bz * BC;
    const int tpl = (g.Wo + LT - 1) / LT;              // tiles per output line
    const int ntiles = g.Do * g.Ho * tpl;
    const int t_begin = split * tiles_per_split;
    const int t_end = t_begin + tiles_per_split < ntiles ? t_begin + tiles_per_split : ntiles;

    const int piece = tid & 15, rsub = tid >> 4;       // 16 rows x 16 pieces per pass
    const int cA = c0 + piece * 4, cD = n0 + piece * 4;
    const bool cokA = cA < g.Cin, cokD = cD < g.Cout;
    float4 tsc = make_float4(1, 1, 1, 1), tsh = make_float4(0, 0, 0, 0);
    if (XF && in_bn && cokA) {
        tsc = *reinterpret_cast<const float4*>(in_bn + cA);
        tsh = *reinterpret_cast<const float4*>(in_bn + g.Cin + cA);
    }
    const float relu_lo = (flags & LISEC_CONV_IN_RELU) ? 0.f : -INFINITY;

    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0};
    float4 ra[PA], rd[PD];
    unsigned amask = 0, dmask = 0;
    int cur_len = 0, nxt_len = 0;

    // tile -> (output line, w0, len, source line offset or -1)
    auto tile_info = [&](int tile, int& w0, int& len, long long& src_line, long long& dy_line) -> bool {
        const int line = tile / tpl;
        w0 = (tile - line * tpl) * LT;
        len = g.Wo - w0 < LT ? g.Wo - w0 : LT;
        const int d = line / g.Ho, h = line - d * g.Ho;
        const int sd = (d << g.ls_d) - g.pd + kd, sh = (h << g.ls_h) - g.ph + kh;
        dy_line = (long long)line * g.Wo;
        src_line = ((long long)sd * g.Hi + sh) * g.Wi;
        return sd >= 0 && sd < g.Di && sh >= 0 && sh < g.Hi;
    };
    auto next_live = [&](int tile) -> int {
        int w0, len; long long a, b;
        while (tile < t_end && !tile_info(tile, w0, len, a, b)) ++tile;
        return tile;
    };
    auto issue = [&](int tile) {
        int w0, len; long long src_line, dy_line;
        (void)tile_info(tile, w0, len, src_line, dy_line);
        nxt_len = len;
        amask = 0;
        dmask = 0;
#pragma unroll
        for (int p = 0; p < PA; ++p) {                 // halo rows: slot s <-> source w = w0 - pw + s
            const int s_ = p * 16 + rsub;
            const int sw = w0 - g.pw + s_;
            const bool ok = cokA && s_ < len + 2 && sw >= 0 && sw < g.Wi;
            const long long off = ok ? (src_line + sw) * g.in_stride + cA : 0;
            ra[p] = *reinterpret_cast<const float4*>(in + off);
            amask |= ok ? (1u << p) : 0u;
        }
#pragma unroll
        for (int p = 0; p < PD; ++p) {
            const int j = p * 16 + rsub;
            const bool ok = cokD && j < len;
            const long long off = ok ? (dy_line + w0 + j) * g.out_stride + cD : 0;
            rd[p] = *reinterpret_cast<const float4*>(dy + off);      // (zeroed in store(): a select here would make the
            dmask |= ok ? (1u << p) : 0u;                            //  wave wait for the load before its MFMAs)
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int s_ = p * 16 + rsub;
            if (s_ < DR + 2) {
                float4 v = ra[p];
                const bool ok = (amask >> p) & 1;
                if (XF) {
                    v.x = ok ? fmaxf(fmaf(v.x, tsc.x, tsh.x), relu_lo) : 0.f;
                    v.y = ok ? fmaxf(fmaf(v.y, tsc.y, tsh.y), relu_lo) : 0.f;
                    v.z = ok ? fmaxf(fmaf(v.z, tsc.z, tsh.z), relu_lo) : 0.f;
                    v.w = ok ? fmaxf(fmaf(v.w, tsc.w, tsh.w), relu_lo) : 0.f;
                } else {
                    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(sA + s_ * BC + piece * 4) = v;
            }
        }
#pragma unroll
        for (int p = 0; p < PD; ++p)
            if (p * 16 + rsub < DR)
                *reinterpret_cast<float4*>(sD + (p * 16 + rsub) * BC + piece * 4) = (dmask >> p) & 1 ? rd[p] : make_float4(0, 0, 0, 0);
        cur_len = nxt_len;
    };

    const float* aCol = sA + (4 * (lane >> 5)) * BC + (wave >> 1) * 32 + (lane & 31);
    const float* dCol = sD + (4 * (lane >> 5)) * BC + (wave & 1) * 32 + (lane & 31);

    int tile = next_live(t_begin);
    if (tile < t_end) { issue(tile); store(); }
    __syncthreads();
    WGRAD_STAMP(1);
    int stamp_tiles = 0;
    while (tile < t_end) {
        ++stamp_tiles;
        const int ntile = next_live(tile + 1);
        if (ntile < t_end) issue(ntile);
        const int nk = (cur_len + 7) >> 3;             // rows beyond len are zero in sD: whole 8-row chunks only
        // software-pipelined by hand: the ten fragment reads of chunk kk + 1 are issued BEFORE the twelve MFMAs of chunk kk
        // (sched_barrier pins that; left alone the reads follow the MFMAs and every chunk waits out the LDS latency)
        float fa[2][6], fb[2][4];
#define LISEC_WG_READ(S_, KK_)                                                                                          \
        do {                                                                                                            \
            const float* ap = aCol + (KK_) * 8 * BC;                                                                    \
            const float* dp = dCol + (KK_) * 8 * BC;                                                                    \
            fb[S_][0] = dp[0]; fb[S_][1] = dp[BC]; fb[S_][2] = dp[2 * BC]; fb[S_][3] = dp[3 * BC];                      \
            fa[S_][0] = ap[0]; fa[S_][1] = ap[BC]; fa[S_][2] = ap[2 * BC]; fa[S_][3] = ap[3 * BC];                      \
            fa[S_][4] = ap[4 * BC]; fa[S_][5] = ap[5 * BC];                                                             \
        } while (0)
#define LISEC_WG_MFMA(S_)                                                                                               \
        do {                                                                                                            \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][0], fb[S_][0], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][1], fb[S_][0], acc1, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][2], fb[S_][0], acc2, 0, 0, 0);                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][1], fb[S_][1], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][2], fb[S_][1], acc1, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][3], fb[S_][1], acc2, 0, 0, 0);                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][2], fb[S_][2], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][3], fb[S_][2], acc1, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][4], fb[S_][2], acc2, 0, 0, 0);                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][3], fb[S_][3], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][4], fb[S_][3], acc1, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][5], fb[S_][3], acc2, 0, 0, 0);                           \
        } while (0)
        int kk = 0;
        if (nk > 0) LISEC_WG_READ(0, 0);
        for (; kk + 1 < nk; kk += 2) {
            LISEC_WG_READ(1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            LISEC_WG_MFMA(0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < nk) LISEC_WG_READ(0, kk + 2);
            __builtin_amdgcn_sched_barrier(0);
            LISEC_WG_MFMA(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kk < nk) LISEC_WG_MFMA(0);
#undef LISEC_WG_READ
#undef LISEC_WG_MFMA
        __syncthreads();
        if (ntile < t_end) store();
        __syncthreads();
        tile = ntile;
    }

    WGRAD_STAMP(2);
    if (stamps && threadIdx.x == 0 && stamp_wg < 8192) stamps[(size_t)stamp_wg * 8 + 5] = (unsigned long long)stamp_tiles;
    const int ntaps = g.KD * g.KH * g.KW;
    const int tap0 = (kd * g.KH + kh) * g.KW;
    const int n = n0 + (wave & 1) * 32 + (lane & 31);
    float* out_base = partial;
    size_t out_split = (size_t)split;
    bool transpose = false;
    if (it.combine) {
        // The slices of a cell (group, c block, n block) meet here, as the K slices of a tile do in igemm.hip: slab stores in
        // the register layout (write-through), every wave waits for its stores, barrier, ONE lane's agent-scope ticket; the
        // last slice to arrive adds the slabs in slice order (deterministic) and writes the finished gradient -- no slab-sum
        // launch.  Used when a cell has few slices (the RPN maps); many slices (the middle layers) keep the slab-sum kernel.
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        // (in the dynamic LDS, free by now: a static word on top of LT = 100's 53 760 B rounds the allocation past a third
        // of the CU's 160 KB and the kernel drops from three to two workgroups per CU)
        volatile int* wg_last_p = reinterpret_cast<volatile int*>(smem);
        const int ncell = ngroups * it.gy * it.gz, cell = (gslot * it.gy + by) * it.gz + bz;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(partial, 0, (int)((size_t)it.nsplit * ncell * kWSlabF4 * 16),
                                                                      0x00020000);
        const unsigned lane_off = (unsigned)((wave * 12) * 64 + lane) * 16u;
        const unsigned mine = (unsigned)(((size_t)split * ncell + cell) * kWSlabF4 * 16) + lane_off;
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const f32x16& a = q < 4 ? acc0 : (q < 8 ? acc1 : acc2);
            const int r = (q & 3) * 4;
            u32x4 v;
            v.x = __float_as_uint(a[r]); v.y = __float_as_uint(a[r + 1]); v.z = __float_as_uint(a[r + 2]); v.w = __float_as_uint(a[r + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine + q * 64 * 16, 0, 16);          // aux 16 = sc1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            *wg_last_p = __hip_atomic_fetch_add(it.counters + cell, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == it.nsplit - 1;
        __syncthreads();
        if (!*wg_last_p) { WGRAD_STAMP(3); return; }
        if (threadIdx.x == 0) __hip_atomic_store(it.counters + cell, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f32x16 s0 = {0}, s1 = {0}, s2 = {0};
        const unsigned zstride = (unsigned)((size_t)ncell * kWSlabF4 * 16);
        unsigned off = (unsigned)((size_t)cell * kWSlabF4 * 16) + lane_off;
        for (int z = 0; z < it.nsplit; ++z, off += zstride) {
            u32x4 v[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + q * 64 * 16, 0, 16);
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                f32x16& a = q < 4 ? s0 : (q < 8 ? s1 : s2);
                const int r = (q & 3) * 4;
                a[r] += __uint_as_float(v[q].x); a[r + 1] += __uint_as_float(v[q].y);
                a[r + 2] += __uint_as_float(v[q].z); a[r + 3] += __uint_as_float(v[q].w);
            }
        }
        acc0 = s0; acc1 = s1; acc2 = s2;
        out_base = it.dW; out_split = 0; transpose = it.transpose != 0;
    }
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        // flip: a stride-1 transposed gather (o + p - k) run as the plain one (o - (K-1-p) + k') with k' = K-1-k
        const int tap = flip ? ((g.KD - 1 - kd) * g.KH + (g.KH - 1 - kh)) * g.KW + (2 - tt) : tap0 + tt;
        float* base = out_base + (out_split * ntaps + tap) * g.Cin * g.Cout;
        const f32x16& acc = tt == 0 ? acc0 : (tt == 1 ? acc1 : acc2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (c < g.Cin && n < g.Cout) {
                if (transpose) base[(size_t)n * g.Cin + c] = acc[r];          // (taps, Cout, Cin): the Conv2DTranspose layout
                else base[(size_t)c * g.Cout + n] = acc[r];
            }
        }
    }
    WGRAD_STAMP(3);
}

template <bool XF, int NP>
__global__ void __launch_bounds__(kThreads, NP == 7 ? 3 : 2)
k_wgrad_halo(WgradItem it) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wgrad_halo_body<XF, NP>(it, blockIdx.x, blockIdx.y, blockIdx.z, smem, g_wgrad_stamps,
                            (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
}

// Several contractions in ONE launch (the stride-1 convolutions of an RPN block: up to five layers whose maps hold 1 250 -
// 20 000 positions and fill a fraction of the chip each): workgroup -> (item, block of that item's grid).
constexpr int kBatchMax = 6;
struct WgradBatch { int n; int first[kBatchMax + 1]; WgradItem item[kBatchMax]; };
template <bool XF, int NP>
__global__ void __launch_bounds__(kThreads, NP == 7 ? 3 : 2)
k_wgrad_halo_batch(WgradBatch b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.first[i + 1]) ++i;
    const WgradItem& it = b.item[i];
    const int local = blockIdx.x - b.first[i];
    const int bx = local % it.gx, by = (local / it.gx) % it.gy, bz = local / (it.gx * it.gy);
    wgrad_halo_body<XF, NP>(it, bx, by, bz, smem, g_wgrad_stamps, blockIdx.x);
}

// ---- ring variant (round 4): 3 x 3 taps in (h, w) at stride 1, dense positions ------------------------------------------
// The halo kernel gives every (kd, kh) group its own workgroups, so each of them re-reads x and dY: nine passes over both
// operands for a Conv3D (PMC: 573 MB per launch of the second middle block against 123 MB algorithmic), and each workgroup
// stages 202 rows for 3 taps.  Here ONE workgroup owns a run of consecutive output lines h0 .. h1 of one column
//   (cell = 64 c x 64 n block, kd, output plane d, w segment)
// and computes all NINE (kh, kw) taps: 12 waves = 3 kh x 4 quadrants of the 64 x 64 cell, three accumulators (kw) per wave --
// the inner loop of the halo kernel, unchanged.  Output line h reads the input lines h - ph + kh; moving to h + 1 needs ONE new
// line, so the three x lines live in a 3-slot LDS ring and every input line and every dY line is staged once per run:
// 202 rows per 9 taps.  LDS (3 (DR + 2) + DR) x 256 B = 108 032 B at 100-row tiles: one of these and one 52 KB workgroup
// of the data-gradient chain per CU (85 + 41 of the 128 granules).  Slabs are written in the accumulator register layout
// [slice][cell][wave 12][q 12][lane 64] float4 and summed in slice order: by the last slice to arrive when a cell has few
// slices (RPN maps), by k_wgrad_ring_reduce otherwise (the middle layers).
constexpr int kRingThreads = 768;
constexpr int kRSlabF4 = 12 * 12 * 64;             // float4 per (slice, cell): 147 456 B

struct RingGeom {
    int Di, Hi, Wi, Do, Ho, Wo, KD;
    int ls_d, pd, ph, pw;
    int Cin, in_stride, Cout, out_stride;
};

struct RingItem {
    RingGeom g;
    const float* in; const float* in_bn; const float* dy;
    float* slabs;            // behind the arrival counters of the workspace
    float* dW;
    int* counters;           // combine: one arrival counter per cell (kd, c block, n block), zero between calls
    int flags, flip, LT, transpose, combine;
    int R;                   // runs per column
    int tpl;                 // tiles (w segments) per output line
    int npairs;              // (kd, d) pairs whose input plane exists
    int gy, gz;              // c blocks, n blocks
    int nblocks;             // workgroups of this item = gy * gz * tpl * R * npairs
    int ncells;              // KD * gy * gz
    int kd_slices[4];        // slices per cell of depth tap kd = (planes d that kd reads) * tpl * R
    unsigned char pair_kd[16], pair_d[16], pair_rank[16];   // rank: index of d among the planes kd reads
};

// (wave, q, lane, j) of the register layout -> (tap, c, n) of the kernel gradient
__device__ __forceinline__ void ring_store_out(const RingItem& it, int kd, int c0, int n0, int wave, int q, int lane, float4 v) {
    const RingGeom& g = it.g;
    const int kh = wave >> 2, quad = wave & 3, tt = q >> 2, i = q & 3;
    const int tap = it.flip ? ((g.KD - 1 - kd) * 3 + (2 - kh)) * 3 + (2 - tt) : (kd * 3 + kh) * 3 + tt;
    const int c = c0 + (quad >> 1) * 32 + 8 * i + 4 * (lane >> 5);
    const int n = n0 + (quad & 1) * 32 + (lane & 31);
    if (n >= g.Cout || c >= g.Cin) return;
    const float e[4] = {v.x, v.y, v.z, v.w};
    if (it.transpose) {                              // (taps, Cout, Cin): the Conv2DTranspose layout
        float* o = it.dW + ((size_t)tap * g.Cout + n) * g.Cin + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c + j < g.Cin) o[j] = e[j];
    } else {
        float* o = it.dW + ((size_t)tap * g.Cin + c) * g.Cout + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c + j < g.Cin) o[(size_t)j * g.Cout] = e[j];
    }
}

template <bool XF, int NP>
__device__ __forceinline__ void
wgrad_ring_body(const RingItem& it, const int bx, float* smem, unsigned long long* stamps, const unsigned stamp_wg) {
    const RingGeom& g = it.g;
    const float* __restrict__ in = it.in;
    const float* __restrict__ dy = it.dy;
    WGRAD_STAMP(0);
    const int LT = it.LT, DR = (LT + 7) & ~7;
    const int SLOT = (DR + 2) * BC;                    // floats per x line slot
    float* sX = smem;                                  // [3][DR + 2][64]
    float* sD = smem + 3 * SLOT;                       // [DR][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, quad = wave & 3;
    // workgroup -> (cell, w segment, run, (kd, d) pair); the pairs of one place are neighbours on one XCD: they share
    // the dY lines (same d) or the x planes
    int id = xcd_remap(bx, it.nblocks);
    const int pair = id % it.npairs; id /= it.npairs;
    const int run = id % it.R; id /= it.R;
    const int wseg = id % it.tpl; id /= it.tpl;
    const int bz = id % it.gz, by = id / it.gz;
    const int kd = it.pair_kd[pair], d = it.pair_d[pair];
    const int h0 = (int)((long long)run * g.Ho / it.R), h1 = (int)((long long)(run + 1) * g.Ho / it.R);
    const int sd = (d << g.ls_d) - g.pd + kd;          // exists: the host lists only such pairs
    const int w0 = wseg * LT;
    const int len = g.Wo - w0 < LT ? g.Wo - w0 : LT;
    const int c0 = by * BC, n0 = bz * BC;
    const int piece = tid & 15, rsub = tid >> 4;       // 48 rows x 16 pieces per pass
    const int cA = c0 + piece * 4, cD = n0 + piece * 4;
    const bool cokA = cA < g.Cin, cokD = cD < g.Cout;
    // the on-load affine of the x operand lives in LDS (behind the dY tile), not in eight registers per lane
    float* sT = sD + DR * BC;                          // [2][64]: scale, shift of this workgroup's 64 input channels
    if (XF && tid < 32) {
        const int cc = c0 + (tid & 15) * 4;
        const bool sh_ = tid >= 16;
        float4 v = sh_ ? make_float4(0, 0, 0, 0) : make_float4(1, 1, 1, 1);
        if (it.in_bn && cc < g.Cin) v = *reinterpret_cast<const float4*>(it.in_bn + (sh_ ? g.Cin : 0) + cc);
        *reinterpret_cast<float4*>(sT + (sh_ ? BC : 0) + (tid & 15) * 4) = v;
    }
    if (XF) __syncthreads();
    const float relu_lo = (it.flags & LISEC_CONV_IN_RELU) ? 0.f : -INFINITY;

    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0};
    // Staging: ONE register array for the two things a step brings in.  Pass p handles row p * 48 + rsub of the virtual
    // image [x line A: DR + 2 rows][second source: an x line (warm-up) or a dY line (DR rows)]: 5 float4 per thread at
    // 100-row tiles.  Buffer loads against a descriptor that covers exactly ONE line (x) / one tile (dY): rows before
    // the line (negative offset), beyond its end, beyond the tile's rows and channel pieces beyond Cin / Cout read as
    // zeros -- no predicates, no masks, no 64-bit lane addresses.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rr[NP];
    const int XR = DR + 2;
    const int strideX4 = g.in_stride * 4, strideD4 = g.out_stride * 4;                   // bytes per row
    const int voffX = cokA ? (w0 - g.pw + rsub) * strideX4 + cA * 4 : 0x40000000;        // (0x40000000: beyond any line)
    const int voffD = cokD ? rsub * strideD4 + cD * 4 : 0x40000000;
    const int recX = ((g.Wi - 1) * g.in_stride + g.Cin) * 4;
    const int recD = ((len - 1) * g.out_stride + g.Cout) * 4;
    auto x_rsrc = [&](int sh) {
        const bool lv = sh >= 0 && sh < g.Hi;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + ((long long)sd * g.Hi + (lv ? sh : 0)) * g.Wi * g.in_stride,
                                                 0, lv ? recX : 0, 0x00020000);
    };
    auto load2 = [&](int shA, bool second_is_x, int lineB) {          // lineB: input line (x) or output line h (dY)
        __amdgpu_buffer_rsrc_t rsA = x_rsrc(shA);
        __amdgpu_buffer_rsrc_t rsB = second_is_x
            ? x_rsrc(lineB)
            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy) + (((long long)d * g.Ho + lineB) * g.Wo + w0) * g.out_stride,
                                                0, recD, 0x00020000);
        const int voffB = second_is_x ? voffX : voffD, strideB4 = second_is_x ? strideX4 : strideD4;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p * 48 + rsub < XR) rr[p] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voffX + p * 48 * strideX4, 0, 0);
            else                    rr[p] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB + (p * 48 - XR) * strideB4, 0, 0);
        }
    };
    auto store2 = [&](int shA, bool second_is_x, int lineB) {
        float4 tsc = make_float4(1, 1, 1, 1), tsh = make_float4(0, 0, 0, 0);
        if (XF) {
            tsc = *reinterpret_cast<const float4*>(sT + piece * 4);
            tsh = *reinterpret_cast<const float4*>(sT + BC + piece * 4);
        }
        float* dA = sX + ((shA + 3) % 3) * SLOT;
        float* dB = second_is_x ? sX + ((lineB + 3) % 3) * SLOT : sD;
        const int rowsB = second_is_x ? XR : DR;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = p * 48 + rsub;
            const bool first = row < XR;
            const int s_ = first ? row : row - XR;
            if (row < XR + rowsB) {
                float4 v = make_float4(__uint_as_float(rr[p].x), __uint_as_float(rr[p].y), __uint_as_float(rr[p].z),
                                       __uint_as_float(rr[p].w));
                if (XF && (first || second_is_x)) {
                    // the affine turns the zeros of the padding into max(shift, 0): the halo rows outside the map are
                    // zeroed again (rows beyond the tile only meet zero dY rows: any finite value does)
                    const int sw = w0 - g.pw + s_;
                    const bool keep = sw >= 0 && sw < g.Wi;
                    v.x = keep ? fmaxf(fmaf(v.x, tsc.x, tsh.x), relu_lo) : 0.f;
                    v.y = keep ? fmaxf(fmaf(v.y, tsc.y, tsh.y), relu_lo) : 0.f;
                    v.z = keep ? fmaxf(fmaf(v.z, tsc.z, tsh.z), relu_lo) : 0.f;
                    v.w = keep ? fmaxf(fmaf(v.w, tsc.w, tsh.w), relu_lo) : 0.f;
                }
                *reinterpret_cast<float4*>((first ? dA : dB) + s_ * BC + piece * 4) = v;
            }
        }
    };

    // warm-up: the two lines below the first output line's newest one, then the first tile
    load2(h0 - g.ph, true, h0 - g.ph + 1);
    store2(h0 - g.ph, true, h0 - g.ph + 1);
    load2(h0 - g.ph + 2, false, h0);
    store2(h0 - g.ph + 2, false, h0);
    __syncthreads();
    WGRAD_STAMP(1);

    const float* dCol = sD + (4 * (lane >> 5)) * BC + (quad & 1) * 32 + (lane & 31);
    const int aoff = (4 * (lane >> 5)) * BC + (quad >> 1) * 32 + (lane & 31);
    const int nk = (len + 7) >> 3;                     // rows beyond len are zero in sD: whole 8-row chunks only
    for (int h = h0; h < h1; ++h) {
        const bool more = h + 1 < h1;
        if (more) load2(h + 3 - g.ph, false, h + 1);
        const int sh = h - g.ph + kh;                  // this wave's input line (wave-uniform)
        if (sh >= 0 && sh < g.Hi) {
            const float* aCol = sX + ((sh + 3) % 3) * SLOT + aoff;
            // software-pipelined by hand as in the halo kernel: the ten fragment reads of chunk kk + 1 are issued BEFORE
            // the twelve MFMAs of chunk kk
            float fa[2][6], fb[2][4];
#define LISEC_WG_READ(S_, KK_)                                                                                          \
            do {                                                                                                        \
                const float* ap = aCol + (KK_) * 8 * BC;                                                                \
                const float* dp = dCol + (KK_) * 8 * BC;                                                                \
                fb[S_][0] = dp[0]; fb[S_][1] = dp[BC]; fb[S_][2] = dp[2 * BC]; fb[S_][3] = dp[3 * BC];                  \
                fa[S_][0] = ap[0]; fa[S_][1] = ap[BC]; fa[S_][2] = ap[2 * BC]; fa[S_][3] = ap[3 * BC];                  \
                fa[S_][4] = ap[4 * BC]; fa[S_][5] = ap[5 * BC];                                                         \
            } while (0)
#define LISEC_WG_MFMA(S_)                                                                                               \
            do {                                                                                                        \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][0], fb[S_][0], acc0, 0, 0, 0);                       \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][1], fb[S_][0], acc1, 0, 0, 0);                       \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][2], fb[S_][0], acc2, 0, 0, 0);                       \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][1], fb[S_][1], acc0, 0, 0, 0);                       \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][2], fb[S_][1], acc1, 0, 0, 0);                       \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][3], fb[S_][1], acc2, 0, 0, 0);                       \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][2], fb[S_][2], acc0, 0, 0, 0);                       \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][3], fb[S_][2], acc1, 0, 0, 0);                       \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][4], fb[S_][2], acc2, 0, 0, 0);                       \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][3], fb[S_][3], acc0, 0, 0, 0);                       \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][4], fb[S_][3], acc1, 0, 0, 0);                       \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[S_][5], fb[S_][3], acc2, 0, 0, 0);                       \
            } while (0)
            int kk = 0;
            if (nk > 0) LISEC_WG_READ(0, 0);
            for (; kk + 1 < nk; kk += 2) {
                LISEC_WG_READ(1, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                LISEC_WG_MFMA(0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < nk) LISEC_WG_READ(0, kk + 2);
                __builtin_amdgcn_sched_barrier(0);
                LISEC_WG_MFMA(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk < nk) LISEC_WG_MFMA(0);
#undef LISEC_WG_READ
#undef LISEC_WG_MFMA
        }
        __syncthreads();
        if (more) store2(h + 3 - g.ph, false, h + 1);                // the slot of line h - ph: no tile reads it again
        __syncthreads();
    }
    WGRAD_STAMP(2);
    if (stamps && threadIdx.x == 0 && stamp_wg < 8192) stamps[(size_t)stamp_wg * 8 + 5] = (unsigned long long)(h1 - h0);

    // slab of this slice, register layout, write-through
    const int cell = (kd * it.gy + by) * it.gz + bz;
    const int slice = (it.pair_rank[pair] * it.tpl + wseg) * it.R + run;
    const int nsl = it.kd_slices[kd];
    int smax = it.kd_slices[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) smax = it.kd_slices[k] > smax ? it.kd_slices[k] : smax;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(it.slabs, 0, (int)((size_t)smax * it.ncells * kRSlabF4 * 16),
                                                                  0x00020000);
    const unsigned lane_off = (unsigned)((wave * 12) * 64 + lane) * 16u;
    const unsigned mine = (unsigned)(((size_t)slice * it.ncells + cell) * kRSlabF4 * 16) + lane_off;
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const f32x16& a = q < 4 ? acc0 : (q < 8 ? acc1 : acc2);
        const int r = (q & 3) * 4;
        u32x4 v;
        v.x = __float_as_uint(a[r]); v.y = __float_as_uint(a[r + 1]); v.z = __float_as_uint(a[r + 2]); v.w = __float_as_uint(a[r + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine + q * 64 * 16, 0, 16);              // aux 16 = sc1
    }
    if (!it.combine) { WGRAD_STAMP(3); return; }
    // few slices per cell: they meet here as the K slices of a tile do in igemm.hip (stores waited for, barrier, one
    // agent-scope ticket); the last one to arrive adds the slabs in slice order and writes the finished gradient
    volatile int* wg_last_p = reinterpret_cast<volatile int*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        *wg_last_p = __hip_atomic_fetch_add(it.counters + cell, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsl - 1;
    __syncthreads();
    if (!*wg_last_p) { WGRAD_STAMP(3); return; }
    if (threadIdx.x == 0) __hip_atomic_store(it.counters + cell, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f32x16 s0 = {0}, s1 = {0}, s2 = {0};
    const unsigned zstride = (unsigned)((size_t)it.ncells * kRSlabF4 * 16);
    unsigned off = (unsigned)((size_t)cell * kRSlabF4 * 16) + lane_off;
    for (int z = 0; z < nsl; ++z, off += zstride) {
        u32x4 v[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) v[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + q * 64 * 16, 0, 16);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            f32x16& a = q < 4 ? s0 : (q < 8 ? s1 : s2);
            const int r = (q & 3) * 4;
            a[r] += __uint_as_float(v[q].x); a[r + 1] += __uint_as_float(v[q].y);
            a[r + 2] += __uint_as_float(v[q].z); a[r + 3] += __uint_as_float(v[q].w);
        }
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const f32x16& a = q < 4 ? s0 : (q < 8 ? s1 : s2);
        const int r = (q & 3) * 4;
        ring_store_out(it, kd, c0, n0, wave, q, lane, make_float4(a[r], a[r + 1], a[r + 2], a[r + 3]));
    }
    WGRAD_STAMP(3);
}

// <= 120 registers: three of these waves and one 148-register wave of the data-gradient chain share a SIMD's 512
template <bool XF, int NP>
__global__ void __launch_bounds__(kRingThreads) __attribute__((amdgpu_num_vgpr(120)))
k_wgrad_ring(RingItem it) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wgrad_ring_body<XF, NP>(it, blockIdx.x, smem, g_wgrad_stamps, blockIdx.x);
}

struct RingBatch { int n; int first[kBatchMax + 1]; RingItem item[kBatchMax]; };
template <bool XF, int NP>
__global__ void __launch_bounds__(kRingThreads) __attribute__((amdgpu_num_vgpr(120)))
k_wgrad_ring_batch(RingBatch b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.first[i + 1]) ++i;
    wgrad_ring_body<XF, NP>(b.item[i], blockIdx.x - b.first[i], smem, g_wgrad_stamps, blockIdx.x);
}

// Slab sum of the ring kernel for cells with many slices (the middle layers: 60 - 80 slices of 147 KB per cell): one WAVE per
// 8 float4 of a cell's slab, lane = 8 * group + t sums the slices group, group + 8, ... of float4 t (four loads in
// flight), then the eight partial sums are added across lanes in a fixed tree; <= 56 registers, no LDS (k_wgrad_reduce_lanes
// says why).  A depth tap no plane reads (kd_slices == 0) gets zeros.
__global__ void __launch_bounds__(256, 8)
k_wgrad_ring_reduce(RingItem it) {
    const int lane = threadIdx.x & 63, tx = lane & 7, ty = lane >> 3;
    const long long i4 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + tx;
    const long long total = (long long)it.ncells * kRSlabF4;
    const long long ii = i4 < total ? i4 : 0;
    const int cell = (int)(ii / kRSlabF4), e = (int)(ii - (long long)cell * kRSlabF4);
    const int kd = cell / (it.gy * it.gz);
    const int nsl = it.kd_slices[kd];
    float4 s = make_float4(0, 0, 0, 0);
    if (i4 < total) {
        const float4* src = reinterpret_cast<const float4*>(it.slabs) + (size_t)cell * kRSlabF4 + e;
        const size_t zs = (size_t)it.ncells * kRSlabF4;
        int k = ty;
        for (; k + 3 * 8 < nsl; k += 4 * 8) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(k + 8 * u) * zs];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsl; k += 8) {
            const float4 v = src[(size_t)k * zs];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        s.x += __shfl_down(s.x, o, 64); s.y += __shfl_down(s.y, o, 64);
        s.z += __shfl_down(s.z, o, 64); s.w += __shfl_down(s.w, o, 64);
    }
    if (ty != 0 || i4 >= total) return;
    const int rem = cell - kd * it.gy * it.gz;
    const int by = rem / it.gz, bz = rem - by * it.gz;
    ring_store_out(it, kd, by * BC, bz * BC, e / 768, (e >> 6) % 12, e & 63, s);
}

// dW = sum over splits (index order).  transpose: write [tap][n][c] (Conv2DTranspose kernels are (kh,kw,out,in)).
// One thread per float4 of the kernel; the S slab loads of a thread are independent (batched 8 at a time) and
// added in slab order, so the result does not depend on the launch geometry.
__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float* __restrict__ partial, int nsplit, int ntaps, int Cin, int Cout,
               int transpose, float* __restrict__ dW, unsigned long long dead_taps) {
    const long long per = (long long)ntaps * Cin * Cout;          // Cout % 4 == 0
    const long long per4 = per >> 2;
    const long long tap4 = ((long long)Cin * Cout) >> 2;
    for (long long i4 = blockIdx.x * 256LL + threadIdx.x; i4 < per4; i4 += (long long)gridDim.x * 256) {
        const float4* src = reinterpret_cast<const float4*>(partial) + i4;
        float4 s = make_float4(0, 0, 0, 0);
        int k = (dead_taps >> (i4 / tap4)) & 1 ? nsplit : 0;      // a tap no workgroup ran: its slabs were never written
        for (; k + 4 <= nsplit; k += 4) {               // (four in flight: <= 56 registers, see k_wgrad_reduce_lanes)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(k + u) * per4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsplit; ++k) {
            const float4 v = src[(size_t)k * per4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const long long i = i4 << 2;
        if (!transpose) {
            reinterpret_cast<float4*>(dW)[i4] = s;
        } else {
            const int n = (int)(i % Cout);
            long long t = i / Cout;
            const int c = (int)(t % Cin);
            const int tap = (int)(t / Cin);
            float* o = dW + ((size_t)tap * Cout + n) * Cin + c;
            o[0] = s.x; o[Cin] = s.y; o[2 * (size_t)Cin] = s.z; o[3 * (size_t)Cin] = s.w;
        }
    }
}

// Many slabs over a small kernel (the 64x64 Dense layers: > 1000 slabs of 16 KB): 32 lanes stride over the slabs of
// 8 float4 outputs per block (loads batched 8 deep), combined in lane order through LDS.
__global__ void __launch_bounds__(256, 8)            // <= 56 registers
k_wgrad_reduce_lanes(const float* __restrict__ partial, int nsplit, int ntaps, int Cin, int Cout,
                     int transpose, float* __restrict__ dW, unsigned long long dead_taps) {
    // One WAVE per 8 outputs (float4 each): lane = 8 * g + t sums slabs g, g + 8, g + 16, ... of output t (eight loads in
    // flight), then the eight partial sums of an output are added across lanes in a fixed tree -- no LDS and no barrier:
    // the pass runs beside data-gradient kernels that hold all but a few KB of every CU's LDS, and with 4 KB of LDS per
    // workgroup only one of its workgroups fitted a CU at a time.  Registers for the same reason: three waves of a
    // 148-register contraction leave 56 registers per SIMD lane; at 64 this kernel's waves only found room when a
    // data-gradient workgroup ENDED (171-226 us in the step for the 47 MB of a middle layer, 12 us alone).
    const long long per4 = ((long long)ntaps * Cin * Cout) >> 2;
    const long long tap4 = ((long long)Cin * Cout) >> 2;
    const int lane = threadIdx.x & 63, tx = lane & 7, ty = lane >> 3;
    const long long i4 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + tx;
    float4 s = make_float4(0, 0, 0, 0);
    const bool live = i4 < per4 && !((dead_taps >> ((i4 < per4 ? i4 : 0) / tap4)) & 1);
    if (live) {
        const float4* src = reinterpret_cast<const float4*>(partial) + i4;
        int k = ty;
        for (; k + 3 * 8 < nsplit; k += 4 * 8) {                // (four in flight, not eight: see the register note below)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(k + 8 * u) * per4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsplit; k += 8) {
            const float4 v = src[(size_t)k * per4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {                 // groups g and g + o/8: ((0+1)+(2+3))+((4+5)+(6+7))
        s.x += __shfl_down(s.x, o, 64); s.y += __shfl_down(s.y, o, 64);
        s.z += __shfl_down(s.z, o, 64); s.w += __shfl_down(s.w, o, 64);
    }
    if (ty != 0 || i4 >= per4) return;
    const long long i = i4 << 2;
    if (!transpose) {
        reinterpret_cast<float4*>(dW)[i4] = s;
    } else {
        const int n = (int)(i % Cout);
        long long t = i / Cout;
        const int c = (int)(t % Cin);
        const int tap = (int)(t / Cin);
        float* o = dW + ((size_t)tap * Cout + n) * Cin + c;
        o[0] = s.x; o[Cin] = s.y; o[2 * (size_t)Cin] = s.z; o[3 * (size_t)Cin] = s.w;
    }
}

struct WgradPlan {
    int TG, ngroups, nsplit, tiles_per_split, ntiles;
    int LT;              // halo kernel: rows per tile
    bool halo;
    bool combine;        // halo kernel: few slices per cell -> the last one to arrive sums the slabs and writes dW
    size_t ws_bytes, slab_bytes;
    LiveGroups live;     // halo kernel: the (kd, kh) groups that get workgroups
};

WgradPlan make_plan(const ConvGeom& g, int mode = 0, bool dy_xf = false, int blocks_target = 0) {
    WgradPlan p;
    p.combine = false;
    p.TG = g.KW <= 4 ? g.KW : 1;
    if (g.KW % p.TG) p.TG = 1;
    p.ngroups = g.KD * g.KH * (g.KW / p.TG);
    p.ntiles = cdiv(g.M, BMW);
    // 3 taps along w at stride 1 over every position of the map: the halo kernel (tiles follow the output lines)
    p.halo = mode == 0 && g.KW == 3 && g.ls_w == 0 && !g.row_coords && !dy_xf && g.Wo >= 8;
    p.live.n = 0;
    if (p.halo) {
        p.TG = 3;
        p.LT = cdiv(g.Wo, cdiv(g.Wo, BMW));         // equal tiles per line
        p.ntiles = g.Do * g.Ho * cdiv(g.Wo, p.LT);
        for (int kd = 0; kd < g.KD; ++kd) {
            bool dlive = false;
            for (int d = 0; d < g.Do && !dlive; ++d) { const int sd = (d << g.ls_d) - g.pd + kd; dlive = sd >= 0 && sd < g.Di; }
            for (int kh = 0; kh < g.KH; ++kh) {
                bool hlive = false;
                for (int h = 0; h < g.Ho && !hlive; ++h) { const int sh = (h << g.ls_h) - g.ph + kh; hlive = sh >= 0 && sh < g.Hi; }
                if (dlive && hlive && p.live.n < 16) p.live.id[p.live.n++] = (unsigned char)(kd * g.KH + kh);
            }
        }
        p.ngroups = p.live.n > 0 ? p.live.n : 1;
    }
    int cb = cdiv(g.Cin, BC), nb = cdiv(g.Cout, BC);
    int base = p.ngroups * cb * nb;
    const int target = blocks_target > 0 ? blocks_target : tuning().wgrad_blocks;
    int want = cdiv(target, base);              // 1024: ~2 rounds of the 512 resident workgroups (measured: 512-1024 blocks
                                                // beat 1536+, whose extra slabs cost more in the reduce than they balance)
    if (want < 1) want = 1;
    if (want > p.ntiles) want = p.ntiles;
    p.tiles_per_split = cdiv(p.ntiles, want);
    p.nsplit = cdiv(p.ntiles, p.tiles_per_split);
    // every plan keeps the head of the workspace (kWgradCounters arrival counters) free, so that the counters of the
    // combining plans stay zero whatever else shares the workspace; p.slab_bytes = what follows the head
    p.slab_bytes = align_up(sizeof(float) * (size_t)p.nsplit * g.KD * g.KH * g.KW * g.Cin * g.Cout, 256);
    p.ws_bytes = sizeof(int) * kWgradCounters + p.slab_bytes;
    if (p.halo && p.nsplit <= tuning().wgrad_combine_max && g.KD * g.KH <= 16 &&
        (long long)g.KD * g.KH * cb * nb <= kWgradCounters) {
        // few slices per cell: combined inside the kernel (every group gets workgroups: the cells of a group that reads
        // nothing must still be written, as zeros)
        p.combine = true;
        p.live.n = g.KD * g.KH;
        for (int i = 0; i < p.live.n; ++i) p.live.id[i] = (unsigned char)i;
        p.ngroups = p.live.n;
        p.slab_bytes = align_up((size_t)p.nsplit * p.ngroups * cb * nb * kWSlabF4 * 16, 256);
        p.ws_bytes = sizeof(int) * kWgradCounters + p.slab_bytes;
    }
    return p;
}

template <int MODE>
int launch_wgrad(const ConvGeom& g, const WgradPlan& p, const float* in, const float* in_bn, int flags,
                 const float* dy, const float* dy_bn, float* partial, hipStream_t st) {
    dim3 grid(p.nsplit * p.ngroups, cdiv(g.Cin, BC), cdiv(g.Cout, BC));
    const size_t lds = 2 * TILE_FLOATS * sizeof(float);
#define LISEC_WG(T)                                                                                               \
    if (g.row_coords)                                                                                             \
        LISEC_LAUNCH((k_wgrad<MODE, T, true>), grid, dim3(kThreads), lds, st, g, in, in_bn, flags, dy, dy_bn, \
                           p.nsplit, p.tiles_per_split, partial);                                                 \
    else                                                                                                          \
        LISEC_LAUNCH((k_wgrad<MODE, T, false>), grid, dim3(kThreads), lds, st, g, in, in_bn, flags, dy, \
                                       dy_bn, p.nsplit, p.tiles_per_split, partial)
    switch (p.TG) {
        case 1: LISEC_WG(1); break;
        case 2: LISEC_WG(2); break;
        case 3: LISEC_WG(3); break;
        default: LISEC_WG(4); break;
    }
#undef LISEC_WG
    LISEC_LAUNCH_CHECK();
    return 0;
}

}  // namespace
}  // namespace lisec

using namespace lisec;

namespace {
// A halo-kernel call made ready: geometry (mirrored if need be), plan, the item the kernel takes.
struct HaloCall { WgradItem it; WgradPlan p; bool xf; int np; unsigned long long dead_taps; };

// returns 0 and fills `hc` when the contraction runs on the halo kernel; 1 when it does not; < 0 on error
int prepare_halo(const lisec_conv_geom* c, const float* in, const float* in_bnstate, int flags, const float* dy,
                 const float* dy_bnstate, int transpose_out, float* dW, bool has_rows, int blocks_target, HaloCall* hc) {
    ConvGeom& g = hc->it.g;
    if (int rc = conv_geom_check(c, &g)) return rc;
    const bool dy_xf = dy_bnstate != nullptr || (flags & LISEC_CONV_DY_RELU);
    // a transposed gather with unit strides is the plain gather with mirrored taps and pads K-1-p: the halo kernel
    // serves it (first deconv: kernel 3, stride 1, 'same')
    const bool flip = c->mode == 1 && g.ls_d == 0 && g.ls_h == 0 && g.ls_w == 0 && g.KW == 3 && !has_rows && !dy_xf;
    if (flip) { g.pd = g.KD - 1 - g.pd; g.ph = g.KH - 1 - g.ph; g.pw = g.KW - 1 - g.pw; }
    if (has_rows) return 1;
    hc->p = make_plan(g, flip ? 0 : c->mode, dy_xf, blocks_target);
    if (!hc->p.halo) return 1;
    const WgradPlan& p = hc->p;
    WgradItem& it = hc->it;
    it.in = in; it.in_bn = in_bnstate; it.dy = dy; it.dW = dW;
    it.flags = flags; it.nsplit = p.nsplit; it.tiles_per_split = p.tiles_per_split; it.flip = flip ? 1 : 0; it.LT = p.LT;
    it.transpose = transpose_out; it.combine = p.combine ? 1 : 0;
    it.gx = p.nsplit * p.ngroups; it.gy = cdiv(g.Cin, BC); it.gz = cdiv(g.Cout, BC);
    it.live = p.live;
    hc->xf = in_bnstate || (flags & LISEC_CONV_IN_RELU);
    hc->np = p.LT + 2 <= 7 * 16 ? 7 : 9;
    hc->dead_taps = 0;
    unsigned ran = 0;
    for (int i = 0; i < p.live.n; ++i) ran |= 1u << p.live.id[i];
    for (int gi = 0; gi < g.KD * g.KH; ++gi) {
        if ((ran >> gi) & 1) continue;
        const int kd = gi / g.KH, kh = gi - kd * g.KH;
        for (int tt = 0; tt < 3; ++tt) {
            const int tap = flip ? ((g.KD - 1 - kd) * g.KH + (g.KH - 1 - kh)) * g.KW + (2 - tt) : gi * g.KW + tt;
            hc->dead_taps |= 1ULL << tap;
        }
    }
    return 0;
}

// workspace = [kWgradCounters arrival counters][slabs]; counter0: first counter of this call, slab_off: bytes into the slabs
void place_workspace(HaloCall* hc, void* workspace, int counter0, size_t slab_off) {
    hc->it.counters = static_cast<int*>(workspace) + counter0;
    hc->it.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + sizeof(int) * kWgradCounters + slab_off);
}

// LDS of a halo workgroup.  At LT = 100 that is 53 760 B = 42 of the CU's 128 granules of 1 280 B, and three fit -- which
// starves the data-gradient chain running beside the weight gradients on the other stream of its 41 granules (measured
// in the step: 227.7 samples/s with three per CU against 232 with two; alone the three-per-CU launch is 5 % faster).
// tuning().wgrad_per_cu == 2 (default) therefore asks for 43 granules: two of these + one chain workgroup = 127 of 128.
size_t halo_lds(const HaloCall& hc) {
    const int DR = (hc.p.LT + 7) & ~7;
    size_t lds = (size_t)(2 * DR + 2) * BC * sizeof(float);
    // just over a third of 160 KB and not more: LDS is handed out in 1 280-byte granules, 2 x 43 granules leave 53 760 B --
    // room for the 52 224 B of a data-gradient workgroup; a request of 55 637 B (44 granules) left 51 200 and locked the
    // chain out of every CU that held two of these
    const size_t third = 54700;
    if (tuning().wgrad_per_cu == 2 && lds < third) lds = third;
    return lds;
}

// ---- ring kernel, host side ------------------------------------------------------------------------------------------
struct RingCall { RingItem it; bool xf; int np; int lines; size_t slab_bytes, ws_bytes; };

int ring_slots() {
    if (tuning().wgrad_ring_slots > 0) return tuning().wgrad_ring_slots;
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return cus;                                 // one ring workgroup per CU
}

// returns 0 and fills `rc` when the contraction runs on the ring kernel; 1 when it does not; < 0 on error.
// slots: workgroups this call should fill (0: every CU)
int prepare_ring(const lisec_conv_geom* c, const float* in, const float* in_bnstate, int flags, const float* dy,
                 const float* dy_bnstate, int transpose_out, float* dW, bool has_rows, int slots, RingCall* rc) {
    if (!tuning().wgrad_ring || has_rows) return 1;
    ConvGeom g;
    if (int e = conv_geom_check(c, &g)) return e;
    const bool dy_xf = dy_bnstate != nullptr || (flags & LISEC_CONV_DY_RELU);
    const bool flip = c->mode == 1 && g.ls_d == 0 && g.ls_h == 0 && g.ls_w == 0 && g.KW == 3 && !dy_xf;
    if (flip) { g.pd = g.KD - 1 - g.pd; g.ph = g.KH - 1 - g.ph; g.pw = g.KW - 1 - g.pw; }
    if ((c->mode != 0 && !flip) || dy_xf || g.KW != 3 || g.KH != 3 || g.ls_w != 0 || g.ls_h != 0 || g.Wo < 8 ||
        g.ph < 0 || g.ph > 2 || g.out_stride % 4 || g.Cout % 4)
        return 1;
    RingItem& it = rc->it;
    it.g = RingGeom{g.Di, g.Hi, g.Wi, g.Do, g.Ho, g.Wo, g.KD, g.ls_d, g.pd, g.ph, g.pw, g.Cin, g.in_stride, g.Cout, g.out_stride};
    it.in = in; it.in_bn = in_bnstate; it.dy = dy; it.dW = dW; it.slabs = nullptr; it.counters = nullptr;
    it.flags = flags; it.flip = flip ? 1 : 0; it.transpose = transpose_out;
    it.LT = cdiv(g.Wo, cdiv(g.Wo, BMW));        // equal tiles per line
    it.tpl = cdiv(g.Wo, it.LT);
    it.gy = cdiv(g.Cin, BC); it.gz = cdiv(g.Cout, BC);
    it.ncells = g.KD * it.gy * it.gz;
    it.npairs = 0;
    int planes[4] = {0, 0, 0, 0};
    for (int kd = 0; kd < g.KD; ++kd)
        for (int d = 0; d < g.Do; ++d) {
            const int sd = (d << g.ls_d) - g.pd + kd;
            if (sd < 0 || sd >= g.Di) continue;
            if (it.npairs >= 16) return 1;
            it.pair_kd[it.npairs] = (unsigned char)kd; it.pair_d[it.npairs] = (unsigned char)d;
            it.pair_rank[it.npairs] = (unsigned char)planes[kd]++;
            ++it.npairs;
        }
    if (it.npairs == 0) return 1;
    for (int i = it.npairs; i < 16; ++i) it.pair_kd[i] = it.pair_d[i] = it.pair_rank[i] = 0;
    // runs per column: one round of workgroups where possible, the longest runs that balance (a run pays two warm-up
    // lines and one slab: ~ one tile)
    const int cols = it.gy * it.gz * it.tpl * it.npairs;
    if (slots <= 0) slots = ring_slots();
    int best_r = 1;
    long long best_cost = -1;
    for (int r = 1; r <= g.Ho; ++r) {
        const long long rounds = cdiv((long long)cols * r, slots);
        const long long cost = rounds * (cdiv(g.Ho, r) + 1);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_r = r; }
    }
    it.R = best_r;
    rc->lines = cdiv(g.Ho, it.R);
    it.nblocks = cols * it.R;
    int smax = 0;
    bool dead = false;
    for (int kd = 0; kd < 4; ++kd) {
        it.kd_slices[kd] = kd < g.KD ? planes[kd] * it.tpl * it.R : 0;
        if (it.kd_slices[kd] > smax) smax = it.kd_slices[kd];
        if (kd < g.KD && planes[kd] == 0) dead = true;
    }
    it.combine = (!dead && smax <= tuning().wgrad_combine_max && it.ncells <= kWgradCounters) ? 1 : 0;
    rc->slab_bytes = align_up((size_t)smax * it.ncells * kRSlabF4 * 16, 256);
    if (rc->slab_bytes >= (1ull << 31)) return 1;
    rc->ws_bytes = sizeof(int) * kWgradCounters + rc->slab_bytes;
    rc->xf = in_bnstate || (flags & LISEC_CONV_IN_RELU);
    const int DR = (it.LT + 7) & ~7;
    rc->np = 2 * (DR + 2) <= 3 * 48 ? 3 : (2 * (DR + 2) <= 5 * 48 ? 5 : 6);   // 48-row passes over [x line][x line | dY line]
    if ((long long)g.Do * g.Ho * g.Wo * g.out_stride >= (1LL << 31)) return 1;   // int32 offsets into dy
    return 0;
}

size_t ring_lds(const RingCall& rc) {
    const int DR = (rc.it.LT + 7) & ~7;
    return (size_t)(3 * (DR + 2) + DR + 2) * BC * sizeof(float);     // x ring, dY tile, scale/shift
}

void place_ring(RingCall* rc, void* workspace, int counter0, size_t slab_off) {
    rc->it.counters = static_cast<int*>(workspace) + counter0;
    rc->it.slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + sizeof(int) * kWgradCounters + slab_off);
}

int launch_ring_reduce(const RingCall& rc, hipStream_t st) {
    const long long total = (long long)rc.it.ncells * kRSlabF4;
    LISEC_LAUNCH(k_wgrad_ring_reduce, dim3(cdiv(total, 32)), dim3(256), 0, st, rc.it);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

int launch_slab_sum(const ConvGeom& g, const WgradPlan& p, const float* partial, int transpose_out, float* dW,
                    unsigned long long dead_taps, hipStream_t st) {
    const int ntaps = g.KD * g.KH * g.KW;
    long long per = (long long)ntaps * g.Cin * g.Cout;
    int gb = cdiv(per / 4, 256);
    if (gb > 4096) gb = 4096;
    // many slabs: 32 lane groups per output sum nsplit / 32 slabs each (one or two memory round trips instead of
    // nsplit / 8 dependent ones -- the pass runs beside the data-gradient chain and every round trip costs microseconds
    // there: 22-151 us in the step for the 47 MB of a middle layer, 12 us alone)
    if (p.nsplit >= 32)
        LISEC_LAUNCH(k_wgrad_reduce_lanes, dim3(cdiv(per / 4, 32)), dim3(256), 0, st, partial, p.nsplit, ntaps,
                           g.Cin, g.Cout, transpose_out, dW, dead_taps);
    else
        LISEC_LAUNCH(k_wgrad_reduce, dim3(gb), dim3(256), 0, st, partial, p.nsplit, ntaps, g.Cin, g.Cout,
                           transpose_out, dW, dead_taps);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
}  // namespace

extern "C" size_t lisec_conv_wgrad_workspace_bytes(const lisec_conv_geom* c, int row_capacity) {
    ConvGeom g;
    if (conv_geom_check(c, &g)) return 0;
    if (row_capacity > 0) g.M = row_capacity;
    // the plans lisec_conv_wgrad may pick for this geometry: as given, role-swapped (dy transformed), mirrored to mode 0
    const size_t a = make_plan(g, c->mode, false).ws_bytes, b = make_plan(g, 1, true).ws_bytes,
                 m0 = make_plan(g, 0, false).ws_bytes;
    size_t w = a > b ? (a > m0 ? a : m0) : (b > m0 ? b : m0);
    RingCall rc;
    if (row_capacity <= 0 && prepare_ring(c, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, false, 0, &rc) == 0 && rc.ws_bytes > w)
        w = rc.ws_bytes;
    return w;
}

extern "C" int lisec_conv_wgrad_plan_query(const lisec_conv_geom* c, int flags, int has_dy_bnstate, int has_row_list,
                                           int row_capacity, lisec_wgrad_plan* out) {
    LISEC_CHECK_ARG(out, "NULL plan");
    ConvGeom g;
    if (int rc = conv_geom_check(c, &g)) return rc;
    static const int32_t dummy[4] = {0, 0, 0, 0};
    if (has_row_list) {
        LISEC_CHECK_ARG(row_capacity > 0, "row list needs a capacity");
        g.row_coords = dummy; g.row_count = dummy + 3; g.M = row_capacity; g.pointwise = 0;
    }
    const bool dy_xf = has_dy_bnstate || (flags & LISEC_CONV_DY_RELU);
    const bool flip = c->mode == 1 && g.ls_d == 0 && g.ls_h == 0 && g.ls_w == 0 && g.KW == 3 && !has_row_list && !dy_xf;
    if (flip) { g.pd = g.KD - 1 - g.pd; g.ph = g.KH - 1 - g.ph; g.pw = g.KW - 1 - g.pw; }
    const WgradPlan p = make_plan(g, flip ? 0 : c->mode, dy_xf);
    out->halo = p.halo ? 1 : 0;
    out->mirrored = flip ? 1 : 0;
    out->taps_per_group = p.TG;
    out->groups = p.ngroups;
    out->tile_rows = p.halo ? p.LT : BMW;
    out->staging_passes = p.halo ? (p.LT + 2 <= 7 * 16 ? 7 : 9) : 8;
    out->tiles = p.ntiles;
    out->slabs = p.nsplit;
    out->tiles_per_slab = p.tiles_per_split;
    out->workgroups = p.nsplit * p.ngroups * cdiv(g.Cin, BC) * cdiv(g.Cout, BC);
    out->lane_reduce = (!p.combine && p.nsplit >= 32) ? 1 : 0;
    out->combine_in_kernel = p.combine ? 1 : 0;
    out->ring = 0; out->runs_per_column = 0; out->lines_per_run = 0;
    RingCall rc;
    static const float dummy_bn[1] = {0.f};
    const int rk = prepare_ring(c, nullptr, nullptr, flags, nullptr, has_dy_bnstate ? dummy_bn : nullptr, 0, nullptr,
                                has_row_list != 0, 0, &rc);
    if (rk < 0) return rk;
    if (rk == 0) {
        out->ring = 1; out->halo = 0;
        out->taps_per_group = 9;
        out->groups = rc.it.npairs;                    // (kd, d) pairs that read an existing plane
        out->tile_rows = rc.it.LT;
        out->staging_passes = rc.np;
        out->tiles = rc.it.g.Do * rc.it.g.Ho * rc.it.tpl;
        out->runs_per_column = rc.it.R;
        out->lines_per_run = rc.lines;
        int smax = 0;
        for (int k = 0; k < 4; ++k) smax = rc.it.kd_slices[k] > smax ? rc.it.kd_slices[k] : smax;
        out->slabs = smax;
        out->tiles_per_slab = rc.lines;
        out->workgroups = rc.it.nblocks;
        out->lane_reduce = rc.it.combine ? 0 : 1;
        out->combine_in_kernel = rc.it.combine;
    }
    return LISEC_OK;
}

extern "C" int lisec_conv_wgrad(const lisec_conv_geom* c, const float* in, const float* in_bnstate, int flags,
                                const float* dy, const float* dy_bnstate, void* workspace,
                                size_t workspace_bytes, int transpose_out, float* dW, const int32_t* row_coords,
                                const int32_t* row_count, int row_capacity, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(in && dy && workspace && dW, "NULL pointer");
    LISEC_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)dy & 15) == 0, "in/dy must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    {
        RingCall rc;
        const int rk = prepare_ring(c, in, in_bnstate, flags, dy, dy_bnstate, transpose_out, dW, row_coords != nullptr, 0, &rc);
        if (rk < 0) return rk;
        if (rk == 0) {
            if (workspace_bytes < rc.ws_bytes) {
                set_error("wgrad workspace too small: %zu < %zu", workspace_bytes, rc.ws_bytes);
                return LISEC_ENOSPC;
            }
            place_ring(&rc, workspace, 0, 0);
            const size_t lds = ring_lds(rc);
#define LISEC_WR(X_, NP_) LISEC_LAUNCH((k_wgrad_ring<X_, NP_>), dim3(rc.it.nblocks), dim3(kRingThreads), lds, st, rc.it)
            if (rc.np == 3)      { if (rc.xf) LISEC_WR(true, 3); else LISEC_WR(false, 3); }
            else if (rc.np == 5) { if (rc.xf) LISEC_WR(true, 5); else LISEC_WR(false, 5); }
            else                 { if (rc.xf) LISEC_WR(true, 6); else LISEC_WR(false, 6); }
#undef LISEC_WR
            LISEC_LAUNCH_CHECK();
            return rc.it.combine ? LISEC_OK : launch_ring_reduce(rc, st);
        }
    }
    HaloCall hc;
    const int kind = prepare_halo(c, in, in_bnstate, flags, dy, dy_bnstate, transpose_out, dW, row_coords != nullptr, 0, &hc);
    if (kind < 0) return kind;
    ConvGeom& g = hc.it.g;                      // checked (and mirrored, for a unit-stride transposed gather) by prepare_halo
    LISEC_CHECK_ARG(g.out_stride % 4 == 0 && g.Cout % 4 == 0, "dY channels/stride must be multiples of 4");
    if (kind == 0) {
        if (workspace_bytes < hc.p.ws_bytes) {
            set_error("wgrad workspace too small: %zu < %zu", workspace_bytes, hc.p.ws_bytes);
            return LISEC_ENOSPC;
        }
        place_workspace(&hc, workspace, 0, 0);
        dim3 grid(hc.it.gx, hc.it.gy, hc.it.gz);
        const size_t lds = halo_lds(hc);
#define LISEC_WH(X_, NP_) LISEC_LAUNCH((k_wgrad_halo<X_, NP_>), grid, dim3(kThreads), lds, st, hc.it)
        if (hc.np == 7) { if (hc.xf) LISEC_WH(true, 7); else LISEC_WH(false, 7); }
        else            { if (hc.xf) LISEC_WH(true, 9); else LISEC_WH(false, 9); }
#undef LISEC_WH
        LISEC_LAUNCH_CHECK();
        if (hc.p.combine) return LISEC_OK;
        return launch_slab_sum(g, hc.p, hc.it.partial, transpose_out, dW, hc.dead_taps, st);
    }
    if (row_coords) {
        LISEC_CHECK_ARG(row_count && row_capacity > 0, "row list needs a device count and a capacity");
        g.row_coords = row_coords; g.row_count = row_count; g.M = row_capacity; g.pointwise = 0;
    }
    const bool dy_xf = dy_bnstate != nullptr || (flags & LISEC_CONV_DY_RELU);
    WgradPlan p = make_plan(g, c->mode, dy_xf);
    if (workspace_bytes < p.ws_bytes) {
        set_error("wgrad workspace too small: %zu < %zu", workspace_bytes, p.ws_bytes);
        return LISEC_ENOSPC;
    }
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + sizeof(int) * kWgradCounters);
    const int rc = c->mode == 0 ? launch_wgrad<0>(g, p, in, in_bnstate, flags, dy, dy_bnstate, partial, st)
                                : launch_wgrad<1>(g, p, in, in_bnstate, flags, dy, dy_bnstate, partial, st);
    if (rc) return rc;
    return launch_slab_sum(g, p, partial, transpose_out, dW, 0, st);
}

// Several weight gradients in one launch (see k_wgrad_halo_batch): every item must take the halo kernel with the same
// staging variant -- the stride-1 3x3 convolutions of one RPN block do -- otherwise the items run one after the other.
// The launch aims for tuning.wgrad_blocks workgroups over ALL items, in proportion to their rows.
namespace {
int plan_batch(const lisec_wgrad_item* items, int n, HaloCall* hcs, size_t* offsets, size_t* total) {
    // first pass: rows of work per item at one slice, to share the block target
    long long work[kBatchMax], all = 0;
    for (int i = 0; i < n; ++i) {
        HaloCall probe;
        const int kind = prepare_halo(items[i].g, items[i].in, items[i].in_bnstate, items[i].flags, items[i].dy, nullptr,
                                      items[i].transpose_out, items[i].dW, false, 0, &probe);
        if (kind != 0) return kind < 0 ? kind : 1;
        const ConvGeom& g = probe.it.g;
        work[i] = (long long)g.Do * g.Ho * g.Wo * g.KD * g.KH * cdiv(g.Cin, BC) * cdiv(g.Cout, BC);
        all += work[i];
    }
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        int target = (int)((double)tuning().wgrad_batch_blocks * (double)work[i] / (double)all + 0.5);
        if (target < 1) target = 1;
        const int kind = prepare_halo(items[i].g, items[i].in, items[i].in_bnstate, items[i].flags, items[i].dy, nullptr,
                                      items[i].transpose_out, items[i].dW, false, target, &hcs[i]);
        if (kind != 0) return kind < 0 ? kind : 1;
        if (hcs[i].xf != hcs[0].xf || hcs[i].np != hcs[0].np) return 1;
        offsets[i] = off;
        off += hcs[i].p.slab_bytes;
    }
    *total = sizeof(int) * kWgradCounters + off;
    int cells = 0;
    for (int i = 0; i < n; ++i) cells += hcs[i].p.ngroups * hcs[i].it.gy * hcs[i].it.gz;
    if (cells > kWgradCounters) return 1;
    return 0;
}
}  // namespace

namespace {
// ring kernel: every item on it with the same staging variant; the CUs are shared in proportion to the items' work
int plan_ring_batch(const lisec_wgrad_item* items, int n, RingCall* rcs, size_t* offsets, size_t* total) {
    if (!tuning().wgrad_ring) return 1;
    long long work[kBatchMax], all = 0;
    for (int i = 0; i < n; ++i) {
        const int kind = prepare_ring(items[i].g, items[i].in, items[i].in_bnstate, items[i].flags, items[i].dy, nullptr,
                                      items[i].transpose_out, items[i].dW, false, 0, &rcs[i]);
        if (kind != 0) return kind < 0 ? kind : 1;
        const RingItem& it = rcs[i].it;
        work[i] = (long long)it.gy * it.gz * it.tpl * it.npairs * it.g.Ho * (it.LT + 8);
        all += work[i];
    }
    size_t off = 0;
    int cells = 0;
    for (int i = 0; i < n; ++i) {
        int slots = (int)((double)ring_slots() * (double)work[i] / (double)all);
        if (slots < 1) slots = 1;
        const int kind = prepare_ring(items[i].g, items[i].in, items[i].in_bnstate, items[i].flags, items[i].dy, nullptr,
                                      items[i].transpose_out, items[i].dW, false, slots, &rcs[i]);
        if (kind != 0) return kind < 0 ? kind : 1;
        if (rcs[i].xf != rcs[0].xf || rcs[i].np != rcs[0].np) return 1;
        offsets[i] = off;
        off += rcs[i].slab_bytes;
        cells += rcs[i].it.ncells;
    }
    if (cells > kWgradCounters || off >= (1ull << 31)) return 1;
    *total = sizeof(int) * kWgradCounters + off;
    return 0;
}
}  // namespace

extern "C" size_t lisec_conv_wgrad_batched_workspace_bytes(const lisec_wgrad_item* items, int n) {
    if (!items || n < 1 || n > kBatchMax) return 0;
    {
        RingCall rcs[kBatchMax];
        size_t offsets[kBatchMax], total = 0;
        if (plan_ring_batch(items, n, rcs, offsets, &total) == 0) return total;
    }
    HaloCall hcs[kBatchMax];
    size_t offsets[kBatchMax], total = 0;
    if (plan_batch(items, n, hcs, offsets, &total) == 0) return total;
    size_t worst = 0;                               // not batchable: the items run one by one in the same workspace
    for (int i = 0; i < n; ++i) {
        const size_t w = lisec_conv_wgrad_workspace_bytes(items[i].g, 0);
        worst = w > worst ? w : worst;
    }
    return worst;
}

extern "C" int lisec_conv_wgrad_batched(const lisec_wgrad_item* items, int n, void* workspace, size_t workspace_bytes,
                                        lisec_stream_t stream_) {
    LISEC_CHECK_ARG(items && n >= 1 && n <= kBatchMax && workspace, "1 .. 6 items and a workspace");
    hipStream_t st = static_cast<hipStream_t>(stream_);
    {
        RingCall rcs[kBatchMax];
        size_t roff[kBatchMax], rtotal = 0;
        const int rk = plan_ring_batch(items, n, rcs, roff, &rtotal);
        if (rk < 0) return rk;
        if (rk == 0) {
            if (workspace_bytes < rtotal) {
                set_error("batched wgrad workspace too small: %zu < %zu", workspace_bytes, rtotal);
                return LISEC_ENOSPC;
            }
            RingBatch b;
            b.n = n;
            size_t lds = 0;
            int blocks = 0, cells = 0;
            for (int i = 0; i < n; ++i) {
                LISEC_CHECK_ARG(items[i].in && items[i].dy && items[i].dW && ((uintptr_t)items[i].in & 15) == 0 &&
                                ((uintptr_t)items[i].dy & 15) == 0, "item %d: NULL or unaligned tensor", i);
                place_ring(&rcs[i], workspace, cells, roff[i]);
                cells += rcs[i].it.ncells;
                b.first[i] = blocks;
                b.item[i] = rcs[i].it;
                blocks += rcs[i].it.nblocks;
                const size_t l = ring_lds(rcs[i]);
                lds = l > lds ? l : lds;
            }
            b.first[n] = blocks;
#define LISEC_WRB(X_, NP_) LISEC_LAUNCH((k_wgrad_ring_batch<X_, NP_>), dim3(blocks), dim3(kRingThreads), lds, st, b)
            if (rcs[0].np == 3)      { if (rcs[0].xf) LISEC_WRB(true, 3); else LISEC_WRB(false, 3); }
            else if (rcs[0].np == 5) { if (rcs[0].xf) LISEC_WRB(true, 5); else LISEC_WRB(false, 5); }
            else                     { if (rcs[0].xf) LISEC_WRB(true, 6); else LISEC_WRB(false, 6); }
#undef LISEC_WRB
            LISEC_LAUNCH_CHECK();
            for (int i = 0; i < n; ++i)
                if (!rcs[i].it.combine)
                    if (int rc = launch_ring_reduce(rcs[i], st)) return rc;
            return LISEC_OK;
        }
    }
    HaloCall hcs[kBatchMax];
    size_t offsets[kBatchMax], total = 0;
    const int kind = plan_batch(items, n, hcs, offsets, &total);
    if (kind < 0) return kind;
    if (kind != 0) {                                // one by one (same stream: they may share the workspace)
        for (int i = 0; i < n; ++i)
            if (int rc = lisec_conv_wgrad(items[i].g, items[i].in, items[i].in_bnstate, items[i].flags, items[i].dy, nullptr,
                                          workspace, workspace_bytes, items[i].transpose_out, items[i].dW, nullptr, nullptr, 0,
                                          stream_))
                return rc;
        return LISEC_OK;
    }
    if (workspace_bytes < total) {
        set_error("batched wgrad workspace too small: %zu < %zu", workspace_bytes, total);
        return LISEC_ENOSPC;
    }
    WgradBatch b;
    b.n = n;
    size_t lds = 0;
    int blocks = 0, cells = 0;
    for (int i = 0; i < n; ++i) {
        LISEC_CHECK_ARG(items[i].in && items[i].dy && items[i].dW && ((uintptr_t)items[i].in & 15) == 0 &&
                        ((uintptr_t)items[i].dy & 15) == 0, "item %d: NULL or unaligned tensor", i);
        place_workspace(&hcs[i], workspace, cells, offsets[i]);
        cells += hcs[i].p.ngroups * hcs[i].it.gy * hcs[i].it.gz;
        b.first[i] = blocks;
        b.item[i] = hcs[i].it;
        blocks += hcs[i].it.gx * hcs[i].it.gy * hcs[i].it.gz;
        const size_t l = halo_lds(hcs[i]);
        lds = l > lds ? l : lds;
    }
    b.first[n] = blocks;
#define LISEC_WB(X_, NP_) LISEC_LAUNCH((k_wgrad_halo_batch<X_, NP_>), dim3(blocks), dim3(kThreads), lds, st, b)
    if (hcs[0].np == 7) { if (hcs[0].xf) LISEC_WB(true, 7); else LISEC_WB(false, 7); }
    else                { if (hcs[0].xf) LISEC_WB(true, 9); else LISEC_WB(false, 9); }
#undef LISEC_WB
    LISEC_LAUNCH_CHECK();
    for (int i = 0; i < n; ++i)
        if (!hcs[i].p.combine)
            if (int rc = launch_slab_sum(hcs[i].it.g, hcs[i].p, hcs[i].it.partial, items[i].transpose_out, items[i].dW,
                                         hcs[i].dead_taps, st))
                return rc;
    return LISEC_OK;
}

// Diagnostic: points the weight-gradient kernels' stamp buffer at `buf` (device, 8192*8 uint64) or NULL.
extern "C" int lisec_debug_wgrad_stamps(unsigned long long* buf) {
    LISEC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_stamps), &buf, sizeof(buf)));
    return LISEC_OK;
}
