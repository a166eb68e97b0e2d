/*
 * lisec_hip.h -- C ABI of the MI355X (gfx950) hot path of Lisec.
 *
 * The reference (bot15498/Lisec) has no FFI: its hot path is Python calling
 * TensorFlow/Keras.  Every entry point below replaces one group of reference
 * statements; the citation is file:line in the reference tree.
 *
 * Conventions
 *   - plain C types only; every pointer marked "dev" is a device (HBM) pointer the
 *     CALLER owns; nothing here allocates or frees device memory;
 *   - every function takes the HIP stream to enqueue on (void* == hipStream_t) and
 *     returns immediately after enqueueing; no host synchronisation inside;
 *   - return value: 0 ok, < 0 error (LISEC_E*), message via lisec_last_error()
 *     (thread local);
 *   - scratch memory comes from the caller: *_workspace_bytes() says how much;
 *   - tensors are channels-last fp32, exactly the Keras layouts of the reference:
 *       activations (D,H,W,C) / (H,W,C); Dense kernel (in,out);
 *       Conv3D kernel (kd,kh,kw,in,out); Conv2D (kh,kw,in,out);
 *       Conv2DTranspose (kh,kw,out,in).
 */
#ifndef LISEC_HIP_H
#define LISEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISEC_OK 0
#define LISEC_EINVAL (-1)   /* bad argument / shape the kernels do not support */
#define LISEC_ENOSPC (-2)   /* workspace or output capacity too small           */
#define LISEC_EHIP (-3)     /* a HIP runtime call failed                        */

typedef void* lisec_stream_t; /* hipStream_t */

const char* lisec_last_error(void);
/* ABI version, bumped on any signature change. */
int lisec_abi_version(void);
/* Fills name (<= cap bytes) with the device's gcnArchName; returns CU count or < 0. */
int lisec_device_info(char* name, int cap);

/* ------------------------------------------------------------------------------------------
 * 1. Voxeliser  -- replaces get_voxel + VFE_preprocessing + sparse.to_dense
 *    (model_training.py:103-152, :279; Predict.py:21-30).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    double xSize, ySize, zSize;            /* Constants.voxelx/y/z                      */
    int maxVoxelX, maxVoxelY, maxVoxelZ;   /* Constants.nx//2, ny//2, nz                */
    int sampleSize;                        /* Constants.maxPoints (T), <= 64            */
} lisec_voxel_cfg;

/* Header written by lisec_voxelize at the start of `info` (device int32[8]). */
enum { LISEC_VI_NVOX = 0, LISEC_VI_NROWS = 1, LISEC_VI_NVALID = 2, LISEC_VI_MAXCOUNT = 3,
       LISEC_VI_OVERFLOW = 4 };

/* Optional per-sweep side output of lisec_voxelize, input of lisec_vfe_forward: int64[LISEC_ROW_STATS_WORDS].
 *   words [0, LISEC_ROW_STATS_MOMENT_WORDS): the first and second moments of the feature rows -- sum of x_k (6) and
 *     of x_j*x_k (21, j <= k) over all kept rows, as order-independent two-limb fixed-point sums in
 *     LISEC_ROW_STATS_REPLICAS replicas ([replica][27][2]).  The first Dense of the VFE stack has no bias and no
 *     activation in front of its BatchNormalization (model_training.py:171-173,184), so the batch statistics of that
 *     layer are a closed form of these moments and of the kernel: the VFE needs no pass over the rows for them.
 *   the remaining words: scratch of lisec_vfe_forward, zeroed by lisec_voxelize and left zeroed by every forward. */
#define LISEC_ROW_STATS_REPLICAS 16
#define LISEC_ROW_STATS_MOMENT_WORDS (LISEC_ROW_STATS_REPLICAS * 27 * 2)
#define LISEC_ROW_STATS_WORDS (LISEC_ROW_STATS_MOMENT_WORDS + 4 * 64 * 2)

size_t lisec_voxelize_workspace_bytes(const lisec_voxel_cfg* cfg, int n_points);

/*
 * points      dev  n_points rows of `point_stride` elements, first three = x, y, z
 *                  (dtype: 0 = float32, 1 = float64; the reference feeds float64,
 *                  model_training.py:93-96)
 * cap_voxels       capacity of the per-voxel outputs (V <= min(n_points, cells))
 * info        dev  int32[8]   see LISEC_VI_*
 * cell_voxel  dev  int32[NZ*NX*NY]  voxel ordinal of every grid cell, -1 if empty
 * coords      dev  int32[cap_voxels*3]  (z, x, y) as model_training.py:148, sorted by cell
 * counts      dev  int32[cap_voxels]    raw in-range point count
 * npts        dev  int32[cap_voxels]    min(count, sampleSize)
 * row_start   dev  int32[cap_voxels+1]  exclusive prefix of npts
 * rows        dev  float32[n_points*6]  compact feature rows [x,y,z,x-cx,y-cy,z-cz]
 *                  (model_training.py:135-140), voxel v owns rows row_start[v]..+npts[v];
 *                  ascending original point index, first sampleSize kept
 * row_point   dev  int32[n_points] or NULL: original point index of every row
 * row_stats   dev  int64[LISEC_ROW_STATS_WORDS] or NULL: see above
 */
int lisec_voxelize(const lisec_voxel_cfg* cfg, const void* points, int dtype, int n_points,
                   int point_stride, void* workspace, size_t workspace_bytes, int cap_voxels,
                   int32_t* info, int32_t* cell_voxel, int32_t* coords, int32_t* counts,
                   int32_t* npts, int32_t* row_start, float* rows, int32_t* row_point,
                   int64_t* row_stats, lisec_stream_t stream);

/* Expands compact rows into the zero padded (V, T, 6) block layout the reference's
 * SparseTensor / dense tensor uses (model_training.py:141-152).  padded: float32[V*T*6]. */
int lisec_voxel_rows_to_padded(const int32_t* info, const int32_t* npts, const int32_t* row_start,
                               const float* rows, int sampleSize, int cap_voxels, float* padded,
                               lisec_stream_t stream);

/* Lidar ingest (rotate_points + translation of combine_lidar_data, model_training.py:65-98):
 * out[i] = R * raw[i, :3] + t in float64; raw: device float32 rows of raw_stride (5 for Lyft .bin files);
 * rotation9: HOST row-major 3x3 rotation matrix of the sensor quaternion, translation3: HOST;
 * out: device float64 (n,3) -- typically a slice of the concatenated cloud handed to lisec_voxelize. */
int lisec_lidar_transform(const float* raw, int n_points, int raw_stride, const double* rotation9,
                          const double* translation3, double* out, lisec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 2. VFE stack (sparse-exact) -- replaces addVFELayer(6,32) + addVFELayer(32,64) + addFCN(64,64)
 *    + MaxPoolingVFELayer(combine=True) (model_training.py:155-186, :231-235, layers :32-61) on the
 *    dense (D,H,W,T,6) input; output is the dense (D,H,W,64) grid the first Conv3D reads (:236).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float* kernel[3];   /* Dense(use_bias=False) kernels (6,16) (32,32) (64,64)  (:184)          */
    const float* gamma[3];    /* BatchNormalization() variables, 16 / 32 / 64 channels (:171)          */
    const float* beta[3];
    float* moving_mean[3];    /* updated in place by a training forward (momentum 0.99)                */
    float* moving_var[3];
} lisec_vfe_params;

/* Size (floats) of the `saved` buffer a forward fills and the backward reads: BN batch statistics
 * and the per-voxel pre-BN max/min of each layer. */
size_t lisec_vfe_saved_floats(int cap_voxels);
/* The same plus the per-row extras a training forward can leave for the tiled (MFMA) backward: the slot of the first
 * row holding each per-voxel maximum / minimum, and the layer-2 pre-BN value of every class row.  n_points = the row
 * capacity (rows <= n_points).  Pass the same n_points to lisec_vfe_forward and LISEC_VFE_BWD_TILED to the backward. */
size_t lisec_vfe_saved_floats_rows(int cap_voxels, int n_points);
/* Offsets (in floats) inside `saved` of the compact per-voxel outputs written by every forward:
 *   VOUT  float[(cap+1)*64]  grid value of voxel v; row V = the constant every empty cell holds
 *   DELTA float[(cap+1)*64]  VOUT[v] - VOUT[V] */
enum { LISEC_VFE_SAVED_VOUT = 0, LISEC_VFE_SAVED_DELTA = 1 };
size_t lisec_vfe_saved_field_offset(int cap_voxels, int field);
size_t lisec_vfe_workspace_bytes(void);

/*
 * info/cell_voxel/npts/row_start/rows: outputs of lisec_voxelize (device).
 * ncells = NZ*NX*NY, T = sampleSize.  training != 0: batch statistics (Keras fit), moving stats
 * updated; training == 0: moving statistics (Keras predict, Predict.py:38).
 * row_stats: the voxeliser's side output (NULL: the moments are summed here, one more launch).
 * n_points: 0, or the row capacity `saved` was sized for with lisec_vfe_saved_floats_rows (training only: the
 *           per-row extras are then written).
 * grid  dev float32[ncells*64]: (D,H,W,64), every cell written (empty cells hold relu(BN3(.)) != 0); NULL = only
 *       the compact per-voxel outputs (VOUT / DELTA in `saved`), the input of lisec_conv_field_forward.
 */
int lisec_vfe_forward(const lisec_vfe_params* p, const int32_t* info, const int32_t* cell_voxel,
                      const int32_t* npts, const int32_t* row_start, const float* rows, int64_t* row_stats,
                      int n_points, int ncells, int T, int cap_voxels, int training, float* saved, void* workspace,
                      size_t workspace_bytes, float* grid, lisec_stream_t stream);

/* Re-materialises the dense grid from the `saved` buffer of the last lisec_vfe_forward (the HBM-bound
 * writer on its own: lets a caller recycle the 164 MB grid buffer, and lets bench.py time the writer). */
int lisec_vfe_grid_from_saved(const int32_t* info, const int32_t* cell_voxel, int ncells, int cap_voxels,
                              const float* saved, float* grid, lisec_stream_t stream);

/* Gradients of the VFE variables (what fit() derives for the layers of :231-235), training-mode BN.
 * dgrid: float32[ncells*64], gradient wrt the grid written by lisec_vfe_forward(training=1);
 * saved: the buffer that forward filled.  Outputs are written (not accumulated). */
typedef struct {
    float* kernel[3];   /* (6,16) (32,32) (64,64) */
    float* gamma[3];
    float* beta[3];
} lisec_vfe_grads;
#define LISEC_VFE_BWD_TILED 1   /* flags: `saved` carries the per-row extras of lisec_vfe_forward(n_points > 0): layers 3
                                   and 2 run on 32-row tiles on the matrix cores instead of row by row per voxel */
size_t lisec_vfe_backward_workspace_bytes(int cap_voxels, int n_points);
/* Pass EITHER dgrid (dense gradient of the grid) OR the compact form the sparse backward of the first middle
 * layer produces: dout_rows float[(cap_voxels+1)*64] with rows [0,V) = gradient at the occupied cells (row V is
 * written here) and g_all float[64] = sum of the grid gradient over ALL cells. */
int lisec_vfe_backward(const lisec_vfe_params* p, const int32_t* info, const int32_t* cell_voxel,
                       const int32_t* npts, const int32_t* row_start, const float* rows, int n_points,
                       int ncells, int T, int cap_voxels, const float* saved, const float* dgrid,
                       float* dout_rows, const float* g_all, const lisec_vfe_grads* grads, int flags, void* workspace,
                       size_t workspace_bytes, lisec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 3. Dense contractions on the fp32 matrix cores (implicit GEMM, no im2col buffer).
 *    One geometry descriptor covers Conv3D (model_training.py:192-193), Conv2D (:202-203),
 *    Conv2DTranspose (:246,:249,:252), Dense on the last axis (:184,:195), the 1x1 heads (:254-255)
 *    and the data gradients of all of them.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int mode;           /* 0: conv,  src = o*stride - pad + k   (ZeroPaddingND + 'valid', cross-correlation)
                           1: transposed conv, src = (o + pad - k)/stride when divisible
                              (Conv2DTranspose forward; data gradient of a mode-0 conv)                */
    int Di, Hi, Wi;     /* spatial dims of the tensor gathered from (2D: Di = 1)                       */
    int Do, Ho, Wo;     /* spatial dims of the tensor written                                          */
    int KD, KH, KW;     /* kernel taps                                                                 */
    int sd, sh, sw;     /* strides: 1, 2 or 4                                                          */
    int pd, ph, pw;     /* ZeroPadding (mode 0) / padding of the conv this is the transpose of (mode 1) */
    int Cin, in_stride;   /* channels contracted over; floats between positions of `in` (>= Cin)       */
    int Cout, out_stride; /* channels produced; floats between positions of `out` (concat: 768)        */
    int ps, ps_channels;  /* "pixel-shuffle" store for a Conv2DTranspose whose kernel == stride (:249,:252),
                             run as a 1x1 conv with Cout = ps*ps*ps_channels: result column tap*ps_channels+n of
                             input position (h,w) is stored at position (h*ps+kh, w*ps+kw), channel n
                             (tap = kh*ps+kw) of a (Ho*ps, Wo*ps) map; bias is indexed by n.  0 = off.       */
} lisec_conv_geom;

#define LISEC_CONV_IN_RELU 1     /* apply ReLU to the gathered input (after the optional affine)        */
#define LISEC_CONV_OUT_RELU 2    /* apply ReLU before the store (Dense(..., 'relu'), :195)              */
#define LISEC_CONV_ACCUMULATE 4  /* out += result (gradient fan-in)                                     */
#define LISEC_CONV_TAG_ROOFLINE 32 /* measurement aid: one un-sliced launch under the symbol k_igemm<0,false,1>
                                     (lisec_conv_forward_winograd: k_wino<false,0,1>), so a rocprofv3 --stats row
                                     shows this layer alone (bench.py roofline)                                    */
#define LISEC_CONV_DY_RELU 8     /* wgrad only: ReLU on the (optionally affine-transformed) dy operand     */

/* Packed weight layout the kernels read: [tap][K/4][N][4] fp32, K and N zero padded to 64.
 * src element (tap, k, n) is read at src[tap*tap_stride + k*k_stride + n*n_stride], so any Keras
 * kernel layout (and its transpose for the data gradient) packs without a host-side copy. */
size_t lisec_conv_packed_floats(int ntaps, int K, int N);
int lisec_conv_pack_weights(const float* src, int ntaps, int K, int N, long long tap_stride,
                            long long k_stride, long long n_stride, float* dst, lisec_stream_t stream);

/* The same repack for many kernels in one launch (the whole network after an optimizer step).  The table
 * lives in DEVICE memory; start = running sum of ntaps*Kp*Np (Kp, Np = K, N rounded up to 64); total = the
 * sum over all entries. */
typedef struct {
    const float* src;
    float* dst;
    long long tap_stride, k_stride, n_stride, start;
    int ntaps, K, N, Kp, Np, pad_;
} lisec_pack_desc;
int lisec_conv_pack_weights_batched(const lisec_pack_desc* device_table, int n, long long total,
                                    lisec_stream_t stream);

/* Number of 128-row tiles = leading dimension of `stats_partials`. */
int lisec_conv_num_mblocks(const lisec_conv_geom* g);

/*
 * out[m, n] = sum_tap sum_c f(in[src(m,tap), c]) * W[tap][c][n] + bias[n]
 *   in_bnstate  NULL, or float[4*Cin] {scale, shift, mean, invstd}: f(x) = x*scale + shift, the
 *               BatchNormalization of the producing layer (:194,:204), then ReLU if LISEC_CONV_IN_RELU
 *               (:206); padding stays exactly zero, as ZeroPadding follows the activation in the graph
 *   bias        NULL or float[Cout]
 *   stats_partials NULL, or double[num_mblocks][2][Cout]: per-tile sum and sum of squares of the
 *               stored values, the input of lisec_bn_finalize (training-mode batch statistics)
 *   row_coords  NULL (dense: every position of the output map), or a ROW LIST: GEMM row m is the output
 *               position (d,h,w) = row_coords[3m..3m+2] (the voxeliser's `coords`), *row_count rows exist
 *               (device int, <= row_capacity) and out row m is written compactly at out + m*out_stride.
 *               Used to evaluate the gradient of the dense VFE grid only at occupied cells.
 *   workspace   NULL, or scratch of lisec_conv_forward_workspace_bytes(g): lets layers with few output
 *               positions (RPN blocks 2-3) be cut into K slices so that they still fill the 256 CUs; the
 *               slices are combined in a fixed order (deterministic).  With a row list the plan is made for
 *               row_capacity rows: size the scratch with lisec_conv_forward_rows_workspace_bytes.
 */
size_t lisec_conv_forward_workspace_bytes(const lisec_conv_geom* g);
size_t lisec_conv_forward_rows_workspace_bytes(const lisec_conv_geom* g, int row_capacity);
/* The workspace of the contraction entry points must be ZERO-FILLED once, before its first use: it starts with the
 * arrival counters of the K slices (the slices of a tile are combined inside the kernel by the last one to arrive, in
 * slice order -- deterministic, no combine launch), which every call leaves at zero.  One workspace serves any number of
 * layers on ONE stream; calls that may overlap on two streams need a workspace each. */
/* lisec_conv_forward with optional extras (NULL fields = off):
 *   out_mask     the output gate of lisec_conv_forward_masked
 *   bwd_y, bwd_bnstate, bwd_relu
 *                the call computes a gradient dA that is about to cross a BatchNormalization(+ReLU) backwards:
 *                bwd_y is that layer's raw output (positions x Cout, row stride Cout), bwd_bnstate its bnstate.
 *                stats_partials (required; lisec_conv_num_mblocks_bwd(g) rows) then receives per tile
 *                (sum dz, sum dz*yhat), dz = dA * (bwd_relu ? bn(y) > 0 : 1), yhat = (y - mean)*invstd -- pass 1 of
 *                lisec_bn_backward, folded into the store; finish with lisec_bn_backward_apply. */
/* Optional destination of the per-tile BatchNormalization sums of a contraction (instead of the stats_partials
 * table + a lisec_bn_finalize / lisec_bn_backward_apply finaliser launch): the sums are added into order-independent
 * fixed-point accumulators and the LAST workgroup of the call finalises them in place -- deterministic, no extra launch.
 *   acc   device, lisec_bn_sink_words(Cout) int64 words, ZERO before the first use; every call leaves it zeroed
 *   kind  LISEC_SINK_FORWARD: (sum y, sum y^2) -> bnstate as lisec_bn_finalize writes it (+ moving statistics when
 *         moving_mean/moving_var are set)
 *         LISEC_SINK_BACKWARD (with bwd_y): (sum dz, sum dz*yhat) -> dgamma, dbeta and coef float[2*Cout] =
 *         (mean dz, mean dz*yhat), the input of lisec_bn_backward_apply_coef */
#define LISEC_SINK_FORWARD 1
#define LISEC_SINK_BACKWARD 2
typedef struct lisec_bn_sink {
    void* acc;
    int kind;
    int unbiased_moving;
    double n_rows;
    const float* gamma;
    const float* beta;
    float* moving_mean;
    float* moving_var;
    float* bnstate;
    float* dgamma;
    float* dbeta;
    float* coef;
} lisec_bn_sink;
size_t lisec_bn_sink_words(int C);

typedef struct lisec_conv_extras {
    const float* out_mask;
    const float* bwd_y;
    const float* bwd_bnstate;
    int bwd_relu;
    const lisec_bn_sink* sink;
    /* optional, row-list calls: int32[2] device words, ZERO before the first use (every call leaves them zero).  With them a
     * row list whose capacity is large (>= 768 tiles of 128 rows) and whose Cout <= 64 runs as resident workgroups that
     * draw their tiles from a counter -- the tiles of a row list do very unequal work (rows beyond the device-side count,
     * depth parity), which one workgroup per tile turns into idle CUs. */
    int32_t* queue;
    /* optional second contraction riding on the stored tile (model_training.py:195, backwards: the gradient this call stores
     * is the one w.r.t. a middle block's Dense(64, relu) output, gated by out_mask; the Dense's own data gradient is one more
     * 64 x 64 contraction of the same rows):  tail_out[m, :] = out[m, :] (as stored) @ tail_w,  tail_w a packed 64 x 64 kernel
     * (lisec_conv_pack_weights), tail_out (positions, 64).  bwd_y / bwd_bnstate / bwd_relu / sink then describe tail_out.
     * Needs the two-line w-halo kernel (3 taps along w, stride 1, Wo >= 126), Cout = out_stride = 64; LISEC_EINVAL otherwise. */
    const float* tail_w;
    float* tail_out;
    /* optional BatchNormalization-backward APPLY ON LOAD (model_training.py:204-206 backwards): `in` is a gradient w.r.t. the
     * output of BatchNormalization(+ReLU) over the raw map in_y (same layout as `in`), whose backward statistics a sink has
     * already finalised: every gathered element becomes  scale * (gate(y) * g - mean(dz) - yhat * mean(dz * yhat))  -- what
     * lisec_bn_backward_apply_coef would have written -- so the data gradient of a layer runs straight behind the one above it.
     * in_fold_bnstate: the layer's bnstate float[4*Cin]; in_fold_coef: float[2*Cin] (lisec_bn_sink.coef); in_fold_relu: gate.
     * Transposed gathers (mode 1) without in_bnstate / LISEC_CONV_IN_RELU only. */
    const float* in_y;
    const float* in_fold_bnstate;
    const float* in_fold_coef;
    int in_fold_relu;
    /* optional weight gradient of the SAME Dense (model_training.py:195 backwards, what fit() derives for its kernel, :299),
     * for the call that is a Dense(64)'s data gradient with backward statistics (1x1x1, 64 -> 64, packed rows, bwd_y + a
     * LISEC_SINK_BACKWARD sink, no in_bnstate): the call already reads both operands of
     *     dW[i][j] = sum_m f(bwd_y[m, i]) * in[m, j],   f = bwd_bnstate's affine (+ ReLU when bwd_relu)
     * -- the Dense's input is the BatchNormalization output whose statistics the call reduces -- so the resident workgroups
     * accumulate it beside the data gradient instead of a second pass over 2 x (positions x 64) floats.
     * dense_dw: lisec_dense_dw_slabs() x 4096 floats, 16-byte aligned, no initialisation needed; lisec_dense_dw_reduce then
     * sums the slabs in index order (deterministic) into dW (64, 64) = (in channel, out channel), the Keras layout.
     * Needs at least lisec_dense_dw_slabs() tiles of 128 rows; LISEC_EINVAL for any other call. */
    float* dense_dw;
} lisec_conv_extras;
int lisec_dense_dw_slabs(void);
int lisec_dense_dw_reduce(const float* slabs, float* dW, lisec_stream_t stream);
int lisec_conv_num_mblocks_bwd(const lisec_conv_geom* g);

/* The launch plan lisec_conv_forward_ex WILL run for these arguments (pointers only matter as NULL / non-NULL, so the
 * query takes flags for them): which kernel, the workgroup shape, the K slicing, the row order.  Tests pin the plans of
 * the Lyft layer geometries with it; nothing is launched. */
enum { LISEC_KERNEL_IGEMM = 0,   /* generic gather, 128 x 64 tile                                             */
       LISEC_KERNEL_HALO2 = 1,   /* w-halo staging, at most two output lines per tile (Wo >= 126)             */
       LISEC_KERNEL_HALO3 = 2,   /* w-halo staging, three lines per tile (64 <= Wo < 126)                     */
       LISEC_KERNEL_DENSE64 = 3, /* resident workgroups, 1x1 64 -> 64                                         */
       LISEC_KERNEL_QUEUE = 4,   /* resident workgroups drawing row-list tiles from a counter                 */
       LISEC_KERNEL_WIDE = 5 };  /* 128 x 128 tiles, 512-thread workgroups, two-line w-halo staging           */
typedef struct lisec_conv_plan {
    int kernel;          /* LISEC_KERNEL_*                                                                    */
    int cols;            /* output columns per workgroup: 64, 32 (instead of two K slices) or 128 (wide tile) */
    int tiles;           /* 128-row tiles of the call (parity-class order pads every class to whole tiles)    */
    int tail_tile0;      /* first tile of the K-sliced tail (0: the whole layer is sliced, == tiles: none)    */
    int k_slices;        /* K slices of the tail, combined in-kernel by the last slice to arrive              */
    int plane_pair;      /* one workgroup per pair of depth planes                                            */
    int parity_classes;  /* rows visited in (h & 1, w & 1) classes (data gradient of a stride-2 Conv2D)       */
    int workgroups, launches;
    int double_buffered; /* the K-sliced launch runs the two-image kernels, one workgroup per CU              */
} lisec_conv_plan;
int lisec_conv_plan_query(const lisec_conv_geom* g, int has_in_bnstate, int flags, const lisec_conv_extras* extras,
                          int has_stats_table, size_t workspace_bytes, int has_row_list, int row_capacity,
                          lisec_conv_plan* plan);
int lisec_conv_forward_ex(const lisec_conv_geom* g, const float* in, const float* packed_w, const float* bias,
                          const float* in_bnstate, int flags, float* out, const lisec_conv_extras* extras,
                          double* stats_partials, void* workspace, size_t workspace_bytes,
                          const int32_t* row_coords, const int32_t* row_count, int row_capacity,
                          lisec_stream_t stream);
/* lisec_conv_forward with an output gate: stored value = out_mask[m][n] > 0 ? value : 0, out_mask laid out like
 * `out` (same stride).  The data gradient of a layer whose consumer-side activation was a ReLU is gated by that
 * activation while it is stored (what lisec_relu_mask does in a separate pass).  out_mask NULL = no gate. */
int lisec_conv_forward_masked(const lisec_conv_geom* g, const float* in, const float* packed_w, const float* bias,
                              const float* in_bnstate, int flags, float* out, const float* out_mask,
                              double* stats_partials, void* workspace, size_t workspace_bytes,
                              const int32_t* row_coords, const int32_t* row_count, int row_capacity,
                              lisec_stream_t stream);
int lisec_conv_forward(const lisec_conv_geom* g, const float* in, const float* packed_w, const float* bias,
                       const float* in_bnstate, int flags, float* out, double* stats_partials,
                       void* workspace, size_t workspace_bytes, const int32_t* row_coords,
                       const int32_t* row_count, int row_capacity, lisec_stream_t stream);

/* Winograd F(2x2, 3x3) form of lisec_conv_forward_ex (csrc/wino.hip) for contractions with 3 x 3 (h, w) taps, stride 1 and
 * padding 1 along h and w, any depth taps / depth stride, both gather modes: the Conv3D middle blocks
 * (model_training.py:193, 237-238), the stride-1 Conv2Ds of the RPN (:210-214) and the data gradients of both.  Same
 * arithmetic contract as lisec_conv_forward_ex -- fp32 operands, fp32 accumulation, in_bnstate / LISEC_CONV_IN_RELU applied
 * on load with exact zero padding, bias, LISEC_CONV_ACCUMULATE / LISEC_CONV_OUT_RELU, extras.out_mask, extras.bwd_y +
 * a LISEC_SINK_BACKWARD sink, or a LISEC_SINK_FORWARD sink, extras.tail_w / tail_out (Cout = out_stride = 64) -- with 4 / 9 of the multiplications: a 2 x 2 block of outputs is
 * A^T [ sum_c (G g G^T) . (B^T d B) ] A.  Results differ from lisec_conv_forward_ex by fp32 rounding only (the transforms add
 * and halve exactly representable values; tests/test_gpu_winograd.py holds both against the fp64 oracle).
 *   lisec_conv_pack_weights_winograd: G g G^T of every (depth tap, k, n) kernel slice, in the LDS image order of the kernel.
 *       src element (tap, k, n), tap = (kd * 3 + kh) * 3 + kw, is read at src[tap*tap_stride + k*k_stride + n*n_stride];
 *       flip_hw != 0 mirrors the (kh, kw) taps -- the data gradient of a stride-1 correlation is the correlation with the
 *       mirrored kernel and K / N swapped (k_stride / n_stride), so g->mode == 1 calls take a kernel packed with flip_hw = 1.
 *       dst: lisec_conv_winograd_packed_floats(KD, K, N) floats, 16-byte aligned.
 *   lisec_conv_winograd_supported: 1 when lisec_conv_forward_winograd serves these arguments (extras: only the fields named
 *       above; statistics only through a sink; Cin % 8 == 0), else 0 -- nothing is launched.
 *   lisec_conv_forward_winograd: no workspace (never K-sliced); LISEC_EINVAL for arguments it does not serve. */
size_t lisec_conv_winograd_packed_floats(int KD, int K, int N);
int lisec_conv_pack_weights_winograd(const float* src, int KD, int K, int N, long long tap_stride, long long k_stride,
                                     long long n_stride, int flip_hw, float* dst, lisec_stream_t stream);
int lisec_conv_winograd_supported(const lisec_conv_geom* g, int has_in_bnstate, int flags, const lisec_conv_extras* extras);
int lisec_conv_forward_winograd(const lisec_conv_geom* g, const float* in, const float* wino_w, const float* bias,
                                const float* in_bnstate, int flags, float* out, const lisec_conv_extras* extras,
                                lisec_stream_t stream);

/* Winograd F(2x2, 3x3) form of lisec_conv_wgrad (csrc/wino_wgrad.hip) for the 64 -> 64 Conv3D blocks with 3 x 3 (h, w) taps, stride 1
 * and padding 1 along h and w (model_training.py:193, 237-238; what fit() derives for their kernels, :299): dU[kd] = sum over
 * tiles of (B^T d B) (x) (A dY A^T) per transform point, then dW = G^T dU G -- 4 / 9 of the multiplications of the direct weight
 * gradient, fp32 operands and accumulation, slabs summed in a fixed order (deterministic).  g: the geometry of the FORWARD layer
 * (mode 0), in = the layer's input (packed rows of 64), dy (packed rows of 64), dW in the Keras layout (taps, 64, 64).  No
 * on-load affine.  workspace: lisec_conv_wgrad_winograd_workspace_bytes(g), no initialisation needed. */
int lisec_conv_wgrad_winograd_supported(const lisec_conv_geom* g);
size_t lisec_conv_wgrad_winograd_workspace_bytes(const lisec_conv_geom* g);
int lisec_conv_wgrad_winograd(const lisec_conv_geom* g, const float* in, const float* dy, void* workspace,
                              size_t workspace_bytes, float* dW, lisec_stream_t stream);

/* Weight gradient of the contraction described by `g` (the geometry of the FORWARD layer):
 *   dW[tap][c][n] = sum_m f(in[src(m,tap), c]) * dy[m, n]      dy: float32, g->out_stride floats per row
 * Written in the Keras kernel layout (taps, Cin, Cout); transpose_out != 0 writes (taps, Cout, Cin),
 * the Conv2DTranspose layout (kh,kw,out,in).  dy_bnstate (optional, float[4*Cout]) applies the same
 * affine (+ReLU with LISEC_CONV_DY_RELU) to the dy operand: the kernel==stride Conv2DTranspose layers
 * swap roles (in = gathered output gradient, dy = the layer's BN+ReLU input).
 * row_coords/row_count/row_capacity (optional): contraction over a ROW LIST instead of every position: row m
 * is position row_coords[3m..], `in` is gathered there, dy row m is read at dy + m*out_stride.
 * Deterministic: partial slabs reduced in index order. */
size_t lisec_conv_wgrad_workspace_bytes(const lisec_conv_geom* g, int row_capacity);
/* The launch plan lisec_conv_wgrad WILL run (nothing is launched): tests pin the plans of the Lyft layer geometries. */
typedef struct lisec_wgrad_plan {
    int halo;            /* w-halo kernel (3 taps along w at stride 1 over every position): tiles follow the output lines */
    int mirrored;        /* a unit-stride transposed gather run as the plain one with mirrored taps                       */
    int taps_per_group;  /* taps that share one staged dy tile                                                           */
    int groups;          /* tap groups that get workgroups (groups no output line can read are left out)                  */
    int tile_rows;       /* rows per staged tile (halo: equal tiles per output line)                                      */
    int staging_passes;  /* halo: 7 (tiles of <= 110 rows, three workgroups per CU) or 9                                  */
    int tiles, slabs, tiles_per_slab;   /* M tiles, partial slabs (= ranges of tiles), tiles per range                   */
    int workgroups;
    int lane_reduce;     /* >= 32 slabs summed by a separate launch: the lane-strided slab sum                            */
    int combine_in_kernel; /* few slabs per cell: summed by the last slice to arrive, no slab-sum launch                  */
    int ring;            /* ring kernel (3 x 3 taps in (h, w) at stride 1): one 768-thread workgroup owns a run of output  */
                         /* lines of one (cell, kd, plane, w segment) column and all nine taps; the three input lines live */
                         /* in an LDS ring, so x and dy are staged once per nine taps.  groups = (kd, plane) pairs,        */
                         /* taps_per_group = 9, staging_passes = 48-row passes (2 or 3), slabs = most slices of a cell     */
    int runs_per_column; /* ring: runs every column of Ho output lines is cut into                                        */
    int lines_per_run;   /* ring: most output lines of a run                                                               */
} lisec_wgrad_plan;
int lisec_conv_wgrad_plan_query(const lisec_conv_geom* g, int flags, int has_dy_bnstate, int has_row_list, int row_capacity,
                                lisec_wgrad_plan* plan);
int lisec_conv_wgrad(const lisec_conv_geom* g, const float* in, const float* in_bnstate, int flags,
                     const float* dy, const float* dy_bnstate, void* workspace, size_t workspace_bytes,
                     int transpose_out, float* dW, const int32_t* row_coords, const int32_t* row_count,
                     int row_capacity, lisec_stream_t stream);
/* The weight-gradient workspace must be ZERO-FILLED once before its first use, like the contraction workspace: contractions
 * whose slabs are combined inside the kernel keep their arrival counters at its head (left at zero by every call).
 *
 * Several weight gradients in ONE launch: the stride-1 3x3 convolutions of an RPN block (:203, three to five layers whose
 * maps hold 1 250 - 20 000 positions) fill a fraction of the chip each and are leaves of the backward pass, so they can wait
 * for each other and share a launch.  Items that cannot share one (not all on the w-halo kernel with the same staging) run
 * one after the other with the same results.  n <= 6. */
typedef struct lisec_wgrad_item {
    const lisec_conv_geom* g;
    const float* in;
    const float* in_bnstate;
    int flags;
    const float* dy;
    int transpose_out;
    float* dW;
} lisec_wgrad_item;
size_t lisec_conv_wgrad_batched_workspace_bytes(const lisec_wgrad_item* items, int n);
int lisec_conv_wgrad_batched(const lisec_wgrad_item* items, int n, void* workspace, size_t workspace_bytes,
                             lisec_stream_t stream);

/* First middle layer, exact sparse backward (see csrc/sparse_grid.hip).
 *   lisec_conv_tap_sums: S[tap][n] = sum of dy[m][n] over the output positions m of the mode-0 conv `g` whose
 *     tap reads inside the input map; S: float[ntaps*Cout].
 *   lisec_const_field_grads: for an input map equal to the constant vector cvec (Cin) everywhere,
 *     g_all[c] = sum_tap sum_n W[tap][c][n]*S[tap][n]  (sum of the data gradient over ALL input positions)
 *     dW[tap][c][n] += cvec[c]*S[tap][n]               (W, dW: Keras layout (taps, Cin, Cout); either output
 *     may be NULL).  cvec_row (optional, device int): the constant is row min(*cvec_row, cvec_row_max) of the
 *     (rows, Cin) table `cvec` -- the voxel count V only exists on the device. */
size_t lisec_conv_tap_sums_workspace_bytes(const lisec_conv_geom* g);
int lisec_conv_tap_sums(const lisec_conv_geom* g, const float* dy, float* S, void* workspace,
                        size_t workspace_bytes, lisec_stream_t stream);
/* lisec_conv_tap_sums fused with the apply pass of the BatchNormalization backward in front of it (what
 * lisec_bn_backward_apply_coef(relu = 0) computes): dz is the gradient of the normalised map, y / bnstate / coef as
 * there; dy = scale * (dz - coef[c] - yhat * coef[C + c]) is stored (dy may alias dz, row stride Cout) and S is
 * summed over dy -- one pass over the map instead of two. */
int lisec_conv_tap_sums_bn(const lisec_conv_geom* g, const float* dz, const float* y, const float* bnstate,
                           const float* coef, float* dy, float* S, void* workspace, size_t workspace_bytes,
                           lisec_stream_t stream);
/* S == NULL in lisec_conv_tap_sums_bn stops after the per-line sums (left in `workspace`); lisec_conv_tap_sums_finish sums
 * them into S later -- on another stream, beside the row-list data gradient that only needs the dy the first pass stored. */
int lisec_conv_tap_sums_finish(const lisec_conv_geom* g, const void* workspace, size_t workspace_bytes, float* S,
                               lisec_stream_t stream);
int lisec_const_field_grads(const float* W, const float* S, const float* cvec, const int32_t* cvec_row,
                            int cvec_row_max, int ntaps, int Cin, int Cout, float* dW, float* g_all,
                            lisec_stream_t stream);

/* First middle layer, forward over the VFE's compact output instead of the dense grid (csrc/field_conv.hip): the tensor
 * Conv3D(64, 3, (2,1,1)) reads at model_training.py:236 is a constant vector on the empty cells plus V voxel rows, so
 *   out[p] = bias + sum_{taps inside the grid} W[tap]^T c + sum_{taps that read voxel v} W[tap]^T delta_v
 * is evaluated as one V x 64 x (27*64) contraction and one pass that writes `out` -- the same values as
 * lisec_conv_forward on the dense grid up to fp32 summation order, ~1 GFLOP instead of 70.8 on a Lyft sweep.
 *   g          mode-0 geometry of the layer, Cin == Cout == 64, at most 3 taps per axis
 *   vout/delta the VOUT / DELTA fields of lisec_vfe_forward's `saved` buffer ((row_capacity+1, 64) each)
 *   info, coords, cell_voxel: the voxeliser's outputs; row_capacity: its voxel capacity
 *   packed_w   the layer's kernel from lisec_conv_pack_weights (the layout lisec_conv_forward takes)
 *   out        float32 (Do*Ho*Wo, out_stride), every row written
 *   sink       optional LISEC_SINK_FORWARD sink: BatchNormalization batch statistics of `out`, finalised in the call
 * Deterministic (every sum in a fixed order, statistics in fixed-point accumulators). */
size_t lisec_conv_field_forward_workspace_bytes(const lisec_conv_geom* g, int row_capacity);
int lisec_conv_field_forward(const lisec_conv_geom* g, const float* vout, const float* delta, const int32_t* info,
                             const int32_t* coords, const int32_t* cell_voxel, int row_capacity,
                             const float* packed_w, const float* bias, float* out, const lisec_bn_sink* sink,
                             void* workspace, size_t workspace_bytes, lisec_stream_t stream);

/* Upsampling branches + heads collapsed into 16-channel contractions (csrc/head_fused.hip).  The three
 * Conv2DTranspose layers (model_training.py:246,249,252) carry a bias but no BatchNormalization and no activation, and the
 * two 1x1 heads (:254-255) read their Concatenate (:253) linearly, so
 *     head[m][j] = b'_j + sum_b sum_{tap,c} x_b[src_b(m,tap)][c] * Wc_b[tap][c][j],
 *     Wc_b[tap][c][j] = sum_n W_b[tap][n][c] * H[256 b + n][j],   b'_j = b_j + sum_b sum_n bias_b[n] * H[256 b + n][j]
 * (W_b: the Keras Conv2DTranspose kernel (kh,kw,out,in); H = [W_cls | W_reg], (768,16)): transposed convolutions to 16
 * channels through lisec_conv_forward instead of three to 256 channels and a 768 -> 16 contraction; the (100,200,768)
 * concat tensor and its gradient are never formed.  Same values up to fp32 summation order.
 *   lisec_head_compose: Wc[tap*out_tap_stride + c*out_c_stride + j] (j < 16) for ONE branch; head_w = the 256 rows of H
 *     that branch feeds ((Cup,16), row stride 16); bias_out[j] = (bias_in ? bias_in[j] : 0) + sum_n up_bias[n]*H[n][j]
 *     (bias_out NULL: skipped; chain the branches through bias_in, starting from the heads' own bias).
 *   lisec_head_compose_backward: from G = dL/dWc (same indexing as Wc; the lisec_conv_wgrad of the 16-channel contraction)
 *     and S[j] = sum_m dhead[m][j]:  d_up_kernel (taps,Cup,Cin), d_up_bias (Cup), d_head_w (Cup,16) of that branch
     (d_head_w includes up_bias[n]*S[j]: the branch carries its bias into the heads).
 *   lisec_head_shuffle: the kernel == stride branches run as 1x1 contractions with columns (tap, j):
 *     T_b[pos][tap*16 + j], pos = (h/ps_b)*(Wo/ps_b) + w/ps_b, tap = (h%ps_b)*ps_b + w%ps_b, over the (Ho,Wo,16) head map.
 *     backward == 0: head[m][j] += sum_b T_b[...];  backward != 0: T_b[...] = head[m][j] (head = dL/dhead, T_b = dL/dT_b).
 *     T, ps: HOST arrays of n_branches (<= 4) device pointers / strides. */
int lisec_head_compose(const float* up_kernel, const float* up_bias, const float* head_w, int taps, int Cin, int Cup,
                       long long out_tap_stride, long long out_c_stride, float* Wc, const float* bias_in, float* bias_out,
                       lisec_stream_t stream);
int lisec_head_compose_backward(const float* G, long long g_tap_stride, long long g_c_stride, const float* up_kernel,
                                const float* up_bias, const float* head_w, const float* S, int taps, int Cin, int Cup,
                                float* d_up_kernel, float* d_up_bias, float* d_head_w, lisec_stream_t stream);
int lisec_head_shuffle(float* head, int Ho, int Wo, int n_branches, float* const* T, const int* ps, int backward,
                       lisec_stream_t stream);

/* BatchNormalization statistics (Keras: axis -1, eps 1e-3, momentum 0.99, biased batch variance).
 * bnstate: float[4*C] {scale = gamma*rsqrt(var+eps), shift = beta - mean*scale, mean, invstd}.
 * finalize: reduces partials in index order, optionally updates the moving statistics in place
 * (NULL to skip); fold: bnstate from the moving statistics (inference, Predict.py:38). */
int lisec_bn_finalize(const double* partials, int nparts, int C, double n_rows, const float* gamma,
                      const float* beta, float* moving_mean, float* moving_var, int unbiased_moving,
                      float* bnstate, lisec_stream_t stream);
int lisec_bn_fold(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                  int C, float* bnstate, lisec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 4. Training-step helpers (what Keras' fit() does around the contractions, model_training.py:295-299)
 * ------------------------------------------------------------------------------------------ */
size_t lisec_eltwise_workspace_bytes(void);

/* Backward of BatchNormalization (training mode) optionally followed by ReLU (:171-173, :194, :204-206).
 *   dA (M rows, da_stride floats apart, C channels): gradient wrt the activation a = [relu](y*scale+shift)
 *   y  (M, C) raw pre-BN values; bnstate from the forward's lisec_bn_finalize
 *   dz = dA * [z > 0];  dbeta = sum dz;  dgamma = sum dz*yhat;
 *   dy = scale * (dz - dbeta/M - yhat*dgamma/M)           (may alias dA when da_stride == C)
 *   dbias (optional): sum over rows of dy -- the gradient of the bias of the conv that produced y */
int lisec_bn_backward(const float* dA, int da_stride, const float* y, const float* bnstate, long long M, int C,
                      int relu, float* dgamma, float* dbeta, float* dbias, float* dy, void* workspace,
                      size_t workspace_bytes, lisec_stream_t stream);
/* Passes 2-3 of lisec_bn_backward for a gradient whose pass-1 partials (sum dz, sum dz*yhat per tile) were produced by
 * lisec_conv_forward_ex: sums `partials` (double[nparts][2][C], index order), writes dgamma / dbeta and
 * dy = scale * (dz - mean(dz) - yhat * mean(dz*yhat)); dy may alias dA. */
int lisec_bn_backward_apply(const float* dA, int da_stride, const float* y, const float* bnstate, long long M, int C,
                            int relu, const double* partials, int nparts, float* dgamma, float* dbeta, float* dy,
                            void* workspace, size_t workspace_bytes, lisec_stream_t stream);

/* The apply pass alone, for coefficients a lisec_bn_sink (LISEC_SINK_BACKWARD) produced: coef float[2*C] =
 * (mean dz, mean dz*yhat); dy = scale * (dz - coef[c] - yhat * coef[C + c]); dy may alias dA. */
int lisec_bn_backward_apply_coef(const float* dA, int da_stride, const float* y, const float* bnstate, long long M, int C,
                                 int relu, const float* coef, float* dy, lisec_stream_t stream);

/* n strided 2-D float copies in one launch: dst[r*dst_stride + c] = src[r*src_stride + c], r < rows, c < cols.
 * `device_table` lives in device memory (the pointers are fixed for the life of a model). */
typedef struct lisec_copy_desc {
    const float* src;
    float* dst;
    int rows, cols;
    long long src_stride, dst_stride;
} lisec_copy_desc;
int lisec_copy2d_batched(const lisec_copy_desc* device_table, int n, lisec_stream_t stream);

/* Permute((2,3,4,1)) + Reshape (model_training.py:242-243): in (D,H,W,C) -> out (H,W,C*D), channel c*D + d; HW = H*W.
 * With the reference's Constants.nz = 8 the depth is 1 there and the fold is a view; other nz need this copy.
 * inverse != 0: out (D,H,W,C) <- in (H,W,C*D), the gradient's way back, stored as 0 where mask (laid out like out,
 * optional) is <= 0 -- the ReLU of the Dense layer that produced the folded tensor. */
int lisec_fold_depth(const float* in, float* out, int D, long long HW, int C, int inverse, const float* mask,
                     lisec_stream_t stream);

/* grad[i] = act[i] > 0 ? grad[i] : 0   (backward of Dense(..., 'relu'), :195) */
int lisec_relu_mask(float* grad, const float* act, long long n, lisec_stream_t stream);

/* out[c] = sum_m x[m*stride + c]  (bias gradients of Conv2DTranspose / head layers); C divides 256, or is a
 * multiple of 256 up to 1024 (the 768-channel concat gradient in one pass) */
int lisec_colsum(const float* x, int stride, long long M, int C, float* out, void* workspace,
                 size_t workspace_bytes, lisec_stream_t stream);

/* head: (M,16) = [classification (2) | regression (14)] maps.  kind 0: loss=['mse','mse'] (:296);
 * kind 1: sigmoid cross-entropy + SmoothL1 (BASELINE config 4).  Writes dhead = grad_scale * dLoss/dhead and
 * loss_out[3] = {total, class, regression}. */
int lisec_rpn_loss(const float* head, const float* y_cls, const float* y_reg, long long M, int kind,
                   float grad_scale, float* dhead, float* loss_out, void* workspace, size_t workspace_bytes,
                   lisec_stream_t stream);

/* optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True) (:295) on the flat parameter buffer:
 * v <- m*v - lr_t*g;  w <- w + m*v - lr_t*g;  lr_t = lr/(1 + decay*iterations) is computed by the caller. */
int lisec_sgd_nesterov_step(float* theta, const float* grad, float* velocity, long long n, float lr_t,
                            float momentum, lisec_stream_t stream);

/* The same update with the iteration count kept on the DEVICE, so that a HIP graph captured over a whole training
 * step can be replayed (kernel arguments are frozen in a graph): state = long long[2] {iterations, 0} (the second
 * word is scratch and must be 0 between calls); the kernel computes lr_t = (float)(lr / (1 + decay*iterations)) in
 * double, exactly what the host computes for lisec_sgd_nesterov_step, and increments iterations once every
 * workgroup has read it. */
int lisec_sgd_nesterov_step_dev(float* theta, const float* grad, float* velocity, long long n, double lr,
                                double decay, float momentum, long long* state, lisec_stream_t stream);
/* The same update for a PART of the variables, ahead of the rest of the step: state[0] (the iteration count) is read and NOT
 * incremented -- the call that ends the step (lisec_sgd_nesterov_step_dev over the remaining variables) does that.  The
 * RPN + head variables (94 % of them, model_training.py:245-255) have final gradients long before the middle layers and
 * the VFE are differentiated: their update runs on the second stream under the rest of the backward pass. */
int lisec_sgd_nesterov_step_dev_part(float* theta, const float* grad, float* velocity, long long n, double lr, double decay,
                                     float momentum, const long long* state, lisec_stream_t stream);

/* x *= s  (gradient averaging after the data-parallel all-reduce) */
int lisec_scale(float* x, long long n, float s, lisec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 4b. Data-parallel gradient exchange.  The reference trains in ONE process
 *     (model.fit, model_training.py:299); here whole samples are sharded over one process per GPU and the
 *     only exchange of a step is the all-reduce of the flat gradient buffer, followed by identical
 *     lisec_sgd_nesterov_step calls on every rank.
 *     lisec_comm_t is an RCCL communicator (ncclComm_t): make one with lisec_comm_unique_id (one rank;
 *     ship the LISEC_COMM_ID_BYTES to the others by any means) + lisec_comm_init (every rank, after
 *     hipSetDevice), or pass a communicator the caller already owns.  RCCL is resolved at run time from the
 *     copy the process has loaded (librccl.so.1).
 *     lisec_allreduce_grads: grad[0..n) <- sum over ranks / world, in place, enqueued on `stream` (RCCL
 *     all-reduce + the scale kernel).  `world` is the divisor (the communicator's size for a mean).  Call it
 *     on sub-ranges of the buffer to overlap the exchange with the rest of the backward pass: the RPN + head
 *     gradients -- the tail of the buffer, 94 % of it -- are final long before the middle/VFE ones.
 * ------------------------------------------------------------------------------------------ */
#define LISEC_COMM_ID_BYTES 128
typedef void* lisec_comm_t; /* ncclComm_t */
int lisec_comm_unique_id(void* id /* LISEC_COMM_ID_BYTES, host */);
int lisec_comm_init(int rank, int world, const void* id, lisec_comm_t* comm);
int lisec_comm_destroy(lisec_comm_t comm);
/* LISEC_OK when RCCL can be loaded in this process.  No collective, no device work: every rank calls it BEFORE the
 * collective lisec_comm_init, so that a rank without RCCL is found while the others can still be told. */
int lisec_comm_probe(void);
/* Number of ranks of the communicator (ncclCommCount). */
int lisec_comm_count(lisec_comm_t comm, int* count);
/* world: 0 = divide by the communicator's own size; > 0 must EQUAL it (a mismatch would mis-scale every gradient and is
 * refused with LISEC_EINVAL); < 0 = an explicit divisor -world. */
int lisec_allreduce_grads(lisec_comm_t comm, float* grad, long long n, int world, lisec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 5. Box geometry either side of the network; float64 like the reference.
 *    Anchor grid of rpnToRegion.py:75-150 / serialize_data.py:201-232: outX x outY cells of vx x vy metres
 *    (nx/2, ny/2, 2*voxelx, 2*voxely), two anchors (l, w, h, yaw) per cell.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int outX, outY;
    double vx, vy;
    double anchors[2][4];
} lisec_rpn_cfg;

/* rpnToRegion (rpnToRegion.py:113-164): decode cls (outX*outY rows of cls_stride floats, 2 used) and reg
 * (rows of reg_stride floats, 14 used) into boxes, then greedy NMS (nonMaxSuppressionFast, :18-72) with the
 * rotated IoU of serialize_data.py:138-178.  The reference calls it with maxBoxes=20, overlapThresh=0.
 * out_boxes: double[(max_boxes+1)*7], out_probs: double[max_boxes+1], out_count: device int32. */
size_t lisec_rpn_to_region_workspace_bytes(const lisec_rpn_cfg* cfg, int max_boxes);
int lisec_rpn_to_region(const lisec_rpn_cfg* cfg, const float* cls, int cls_stride, const float* reg,
                        int reg_stride, double overlap_thresh, int max_boxes, void* workspace,
                        size_t workspace_bytes, double* out_boxes, double* out_probs, int32_t* out_count,
                        lisec_stream_t stream);

/* The decode alone (rpnToRegion.py:118-158: anchor grid + applyRegrssion, rows anchor-major as the reference's reshape at
 * :146-148): boxes double[2*outX*outY*7], probs double[2*outX*outY], legal int32[2*outX*outY] (0 where :155-158 removes
 * the box).  What lisec_rpn_to_region runs first; exposed so that it can be checked against the reference's own decode. */
int lisec_rpn_decode(const lisec_rpn_cfg* cfg, const float* cls, int cls_stride, const float* reg, int reg_stride,
                     double* boxes, double* probs, int32_t* legal, lisec_stream_t stream);
/* The box geometry the kernels use, for parity checks: corners double[n*4*2] = boxToShapely's [topRight, botRight, botLeft,
 * topLeft] (serialize_data.py:149-162) of boxes double[n*7]; per pair (2k, 2k+1): pair_out double[(n/2)*4] =
 * {calculateIntersection volume (:140-147), calculateUnion (:165-168), IoU (:178), this library's own polygon area},
 * the first three with the polygon area TAKEN from pair_area[k] when it is not NULL (the reference delegates that area to
 * shapely). */
int lisec_box_geometry(const double* boxes, int n, const double* pair_area, double* corners, double* pair_out,
                       lisec_stream_t stream);

/* preprocessLabels up to the class/regress maps before balancing (serialize_data.py:194-307).
 * fixed_boxes: device double[n_boxes*7], ALREADY scaled by fixBoxScaling (:181-191).
 * valid/overlap: double[outX*outY*2], out_regress: double[outX*outY*14] (all overwritten), indexed with the
 * reference's wrap-around of negative voxel indices. */
size_t lisec_rpn_labels_workspace_bytes(int n_boxes);
int lisec_rpn_labels(const lisec_rpn_cfg* cfg, const double* fixed_boxes, int n_boxes, double iou_lo, double iou_hi,
                     void* workspace, size_t workspace_bytes, double* valid, double* overlap, double* out_regress,
                     lisec_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * 5b. Step plans: a whole training step recorded once and re-issued by ONE call (csrc/plan.hip).
 *     The reference repeats one static schedule 180 times (model.fit(batch_size=1, steps_per_epoch=180),
 *     model_training.py:299).  Between lisec_step_plan_begin and lisec_step_plan_end every launch the CALLING THREAD makes
 *     through this library -- on any stream -- and every lisec_event_record / lisec_stream_wait_event is executed as usual
 *     AND appended to the plan with a copy of its arguments; lisec_step_plan_run re-issues them in the recorded order on
 *     the recorded streams (plain launches, no HIP graph).  Arguments are frozen at record time: what changes from step
 *     to step must sit in device memory the recorded launches point at (the sweep in a fixed-capacity point buffer --
 *     pad it with points outside the grid --, the targets, the device-side iteration counter of
 *     lisec_sgd_nesterov_step_dev).  The caller keeps every buffer alive and unchanged in address for the plan's life.
 * ------------------------------------------------------------------------------------------ */
typedef void* lisec_step_plan_t;
int lisec_step_plan_create(lisec_step_plan_t* plan);
int lisec_step_plan_begin(lisec_step_plan_t plan);
int lisec_step_plan_end(lisec_step_plan_t plan);
int lisec_step_plan_run(lisec_step_plan_t plan);
int lisec_step_plan_size(lisec_step_plan_t plan);    /* recorded operations, -1 for NULL */
int lisec_step_plan_recording(void);                  /* 1 while the calling thread records a plan */
int lisec_step_plan_destroy(lisec_step_plan_t plan);
/* hipEventRecord(event, stream) / hipStreamWaitEvent(stream, event, 0) through the library, so that a recording plan sees
 * the fork / join edges between the streams of a step; event: a hipEvent_t the caller made. */
int lisec_event_record(void* event, lisec_stream_t stream);
int lisec_stream_wait_event(lisec_stream_t stream, void* event);
/* A host function as a step of the plan: fn(arg) runs now and, when the calling thread is recording, again at this place of
 * the sequence in every lisec_step_plan_run (0 = success).  For work between a step's launches that is not a launch of
 * this library -- the data-parallel gradient exchange when it goes through torch.distributed (gloo) instead of
 * lisec_allreduce_grads, which records itself. */
int lisec_step_plan_host_call(int (*fn)(void*), void* arg);

/* ------------------------------------------------------------------------------------------
 * 6. Launch-plan tuning and diagnostics.  Not needed by a caller of the hot path: the defaults are the measured
 *    optimum on MI355X; tools/ and bench.py use them to measure the alternatives.
 *    The tuning record is process-wide and read when a call is planned: change it only while no call is in flight.
 * ------------------------------------------------------------------------------------------ */
typedef struct lisec_tuning {
    int struct_bytes;       /* sizeof(lisec_tuning) of the caller (ABI evolution)                              */
    int max_splitk;         /* most K slices per tile                                              (12)        */
    int splitk_min_steps;   /* least 64-deep K steps per slice                                     (3)         */
    int min_splitk;         /* layers that would get fewer slices run unsliced                     (2)         */
    int plane_pair;         /* pair depth planes that run different numbers of taps                (1)         */
    int dense64;            /* resident-workgroup kernel for Dense(64)                             (1)         */
    int half_n;             /* 32-column workgroups instead of two K slices                        (1)         */
    int vfe_shape;          /* launch shape of the VFE stage kernels, -1 = by capacity             (-1)        */
    int field_seg;          /* segment length of the field combine pass, 0 = by capacity           (0)         */
    int field_tpw;          /* taps per workgroup of the field contraction, 0 = by capacity        (0)         */
    int wgrad_blocks;       /* workgroups a weight-gradient launch aims for                        (1024)      */
    int debug_sync;         /* lisec_vfe_backward synchronises and reports after every launch      (0)         */
    int force_splitk;       /* > 0: every sliceable layer gets exactly this many K slices          (0)         */
    int wgrad_combine_max;  /* weight gradients with at most this many slabs per cell sum them in-kernel (32)   */
    int wgrad_batch_blocks; /* workgroups a batched weight-gradient launch aims for                (1024)      */
    int lone_db;            /* small K-sliced layers: one workgroup per CU on the two-image kernels (1)         */
    int wgrad_per_cu;       /* weight-gradient workgroups per CU: 2 leaves 53 KB of LDS for a workgroup of the   */
                            /* data-gradient chain on the other stream (default), 3 = as many as fit            */
    int wgrad_ring;         /* ring kernel for the 3 x 3 stride-1 weight gradients                 (1)         */
    int wgrad_ring_slots;   /* workgroups a ring launch fills, 0 = one per CU                      (0)         */
    int wide_tile;          /* 128 x 128 tiles (512-thread workgroups) for 128-channel w-halo layers of few tiles (0:  */
                            /* measured equal to the 128 x 32 plan alone, 0.8 % slower in the step -- kept for study)   */
} lisec_tuning;
int lisec_tuning_get(lisec_tuning* t);        /* fills *t with the current record (t->struct_bytes set)          */
int lisec_tuning_set(const lisec_tuning* t);  /* t->struct_bytes must be sizeof(lisec_tuning); every field is range-checked */
/* STATE OF THE LIBRARY BEYOND ITS ARGUMENTS -- all of it:
 *   1. the tuning record above: ONE per process, unsynchronised; read when a call is PLANNED (so a recorded step plan keeps
 *      the launches it recorded: re-record it after lisec_tuning_set);
 *   2. step-plan recording (5b): per THREAD -- between lisec_step_plan_begin and _end every launch, event edge, host call and
 *      lisec_allreduce_grads of that thread is also appended to the plan;
 *   3. lisec_last_error(): per thread;
 *   4. the lisec_debug_*_stamps pointers: __device__ variables of the code object, NULL unless a diagnostic tool sets them.
 * Everything else -- tensors, workspaces, counters, statistics sinks, communicators, events, streams -- belongs to the caller
 * and is passed in. */
/* Zero-fills a workspace whose head holds arrival counters (the K-sliced contractions of lisec_conv_forward*, every
 * lisec_conv_wgrad*) before its FIRST use; every call leaves those counters at zero again.  Non-zero counters (scratch that
 * was never cleared, a call that was interrupted, two streams sharing one workspace) make slices wait for arrivals that
 * never come: tiles stay unwritten, with no error. */
int lisec_workspace_init(void* workspace, size_t bytes, lisec_stream_t stream);

/* 100 MHz s_memrealtime stamps of thread 0 of every workgroup at the kernels' phase boundaries (tools: igemm_stamps.py, wgrad_stamps.py, vfe_stamps.py, field_stamps.py).
 * buf: device uint64[8192 x 8] (igemm); see the tools for the others; NULL (the default) turns them off -- no stamp
 * instruction executes then.  The pointer is a __device__ variable of the code object: set it from one thread. */
int lisec_debug_igemm_stamps(unsigned long long* buf);
int lisec_debug_wgrad_stamps(unsigned long long* buf);
int lisec_debug_vfe_stamps(unsigned long long* buf);
int lisec_debug_field_stamps(unsigned long long* buf);
int lisec_debug_wino_stamps(unsigned long long* buf);

#ifdef __cplusplus
}
#endif
#endif /* LISEC_HIP_H */
