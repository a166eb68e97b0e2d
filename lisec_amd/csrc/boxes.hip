// Box geometry around the hot path (gfx950): RPN target generation and RPN decode + NMS.
//
//   label generation   preprocessLabels            serialize_data.py:194-338  
//   decode + NMS       rpnToRegion / nonMaxSuppressionFast / applyRegrssion   rpnToRegion.py:18-164 (8f2)
//   rotated IoU        boxToShapely / calculateIoU  serialize_data.py:138-178
// The reference evaluates anchors x boxes (2 x 100 x 200 x B) rotated-rectangle intersections one by one
// through shapely (minutes per sample, hence its 6-thread variant).  Here every anchor is a thread, the
// rectangle intersection is Sutherland-Hodgman clipping in registers, all in float64 like the reference.
// Integer/latency-bound work: no MFMA, no LDS beyond the small reductions.
#include "common.h"

namespace lisec {
namespace {

struct Pt { double x, y; };

__device__ __forceinline__ void box_corners(const double* b, Pt* c) {
    // boxToShapely (serialize_data.py:149-162): [topRight, botRight, botLeft, topLeft]
    const double th = b[6], l = b[3], w = b[4];
    const double cs = cos(th), sn = sin(th);
    const double rx = b[0] + cs * (w / 2), ry = b[1] - sn * (w / 2);
    const double lx = b[0] - cs * (w / 2), ly = b[1] + sn * (w / 2);
    const double sx = sn * (l / 2), sy = cs * (l / 2);
    c[0] = {rx + sx, ry + sy}; c[1] = {rx - sx, ry - sy}; c[2] = {lx - sx, ly - sy}; c[3] = {lx + sx, ly + sy};
}

__device__ __forceinline__ double signed_area(const Pt* p, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const Pt& u = p[i];
        const Pt& v = p[i + 1 == n ? 0 : i + 1];
        a += u.x * v.y - v.x * u.y;
    }
    return 0.5 * a;
}

// area of the intersection of two convex quadrilaterals (Sutherland-Hodgman + shoelace)
__device__ double quad_intersection_area(const Pt* pa, const Pt* qa) {
    Pt p[4], q[4];
    const bool pf = signed_area(pa, 4) < 0, qf = signed_area(qa, 4) < 0;
    for (int i = 0; i < 4; ++i) { p[i] = pa[pf ? 3 - i : i]; q[i] = qa[qf ? 3 - i : i]; }
    Pt bufA[10], bufB[10];
    Pt* in = bufA;
    Pt* out = bufB;
    int n = 4;
    for (int i = 0; i < 4; ++i) out[i] = p[i];
    for (int e = 0; e < 4 && n > 0; ++e) {
        Pt* t = in; in = out; out = t;
        const int nin = n;
        n = 0;
        const Pt a = q[e], b = q[(e + 1) & 3];
        const double ex = b.x - a.x, ey = b.y - a.y;
        for (int j = 0; j < nin; ++j) {
            const Pt c = in[j], d = in[j + 1 == nin ? 0 : j + 1];
            const double sc = ex * (c.y - a.y) - ey * (c.x - a.x);
            const double sd = ex * (d.y - a.y) - ey * (d.x - a.x);
            if (sc >= 0) out[n++] = c;
            if ((sc >= 0) != (sd >= 0)) {
                const double t = sc / (sc - sd);
                out[n++] = {c.x + t * (d.x - c.x), c.y + t * (d.y - c.y)};
            }
        }
    }
    if (n < 3) return 0.0;
    return fabs(signed_area(out, n));
}

// calculateIoU (serialize_data.py:170-178): z overlap with the FULL height as half extent, not clamped
__device__ double rot_iou(const double* b1, const double* b2) {
    // rectangles further apart than the sum of their circumradii: polygon area 0 -> IoU exactly 0
    const double dx = b1[0] - b2[0], dy = b1[1] - b2[1];
    const double r = 0.5 * (hypot(b1[3], b1[4]) + hypot(b2[3], b2[4]));
    if (dx * dx + dy * dy > r * r * 1.0000001) return 0.0;
    Pt c1[4], c2[4];
    box_corners(b1, c1);
    box_corners(b2, c2);
    const double area = quad_intersection_area(c1, c2);
    const double botZ = fmax(b1[2] - b1[5], b2[2] - b2[5]);
    const double topZ = fmin(b1[2] + b1[5], b2[2] + b2[5]);
    const double inter = (topZ - botZ) * area;
    const double uni = b1[3] * b1[4] * b1[5] + b2[3] * b2[4] * b2[5] - inter;
    return inter / uni;
}

// The device box geometry, exposed for parity checks against reference-run goldens (tests/golden/box_geometry_decode.npz):
// per box the four corners of boxToShapely (serialize_data.py:149-162); per PAIR (2k, 2k + 1), with the polygon area
// GIVEN (the reference delegates it to shapely), calculateIntersection's volume (:140-147), calculateUnion (:165-168) and
// their quotient (:178) -- and, on the side, the area this file's own clipping finds for the pair.
__global__ void k_box_geometry(const double* __restrict__ boxes, int n, const double* __restrict__ pair_area,
                               double* __restrict__ corners, double* __restrict__ pair_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pt c[4];
    box_corners(boxes + (size_t)i * 7, c);
    for (int k = 0; k < 4; ++k) { corners[((size_t)i * 4 + k) * 2] = c[k].x; corners[((size_t)i * 4 + k) * 2 + 1] = c[k].y; }
    if ((i & 1) || i + 1 >= n) return;
    const double* b1 = boxes + (size_t)i * 7;
    const double* b2 = b1 + 7;
    Pt c2[4];
    box_corners(b2, c2);
    const double area = pair_area ? pair_area[i >> 1] : quad_intersection_area(c, c2);
    const double botZ = fmax(b1[2] - b1[5], b2[2] - b2[5]);
    const double topZ = fmin(b1[2] + b1[5], b2[2] + b2[5]);
    const double inter = (topZ - botZ) * area;
    const double uni = b1[3] * b1[4] * b1[5] + b2[3] * b2[4] * b2[5] - inter;
    double* o = pair_out + (size_t)(i >> 1) * 4;
    o[0] = inter; o[1] = uni; o[2] = inter / uni; o[3] = quad_intersection_area(c, c2);
}

// ---- decode: anchor grid + applyRegrssion (rpnToRegion.py:75-150) ----------------------------------------
__global__ void k_rpn_decode(const float* __restrict__ cls, int cls_stride, const float* __restrict__ reg,
                             int reg_stride, lisec_rpn_cfg cfg, double* __restrict__ boxes,
                             double* __restrict__ probs, int* __restrict__ alive) {
    const int M = cfg.outX * cfg.outY;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * M) return;
    const int a = i / M, m = i - a * M;
    const int ix = m / cfg.outY, iy = m - ix * cfg.outY;
    const double* an = cfg.anchors[a];
    const float* t = reg + (size_t)m * reg_stride + a * 7;
    const double x = ix * cfg.vx + cfg.vx / 2, y = iy * cfg.vy + cfg.vy / 2;
    double* b = boxes + (size_t)i * 7;
    b[0] = (double)t[0] * an[0] + x;
    b[1] = (double)t[1] * an[1] + y;
    b[2] = (double)t[2] * an[2] + 1.0;
    b[3] = exp((double)t[3]) * an[0];
    b[4] = exp((double)t[4]) * an[1];
    b[5] = exp((double)t[5]) * an[2];
    b[6] = (double)t[6] + an[3];
    probs[i] = (double)cls[(size_t)m * cls_stride + a];
    alive[i] = !(b[3] < 0 || b[4] < 0 || b[5] < 0);          // "remove illegal boxes" (rpnToRegion.py:155-158)
}

// highest probability among the live boxes (ties: larger flat index), appended to the pick list
__global__ void __launch_bounds__(1024)
k_nms_pick(const double* __restrict__ probs, int* __restrict__ alive, int n, int* __restrict__ picks,
           int* __restrict__ npicks, int max_picks) {
    __shared__ double sp[1024];
    __shared__ int si[1024];
    double bp = -INFINITY;
    int bi = -1;
    if (*npicks >= max_picks) { return; }
    for (int j = threadIdx.x; j < n; j += 1024) {
        if (alive[j] && (bi < 0 || probs[j] > bp || (probs[j] == bp && j > bi))) { bp = probs[j]; bi = j; }
    }
    sp[threadIdx.x] = bp; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const int oi = si[threadIdx.x + o];
            const double op = sp[threadIdx.x + o];
            if (oi >= 0 && (si[threadIdx.x] < 0 || op > sp[threadIdx.x] || (op == sp[threadIdx.x] && oi > si[threadIdx.x]))) {
                sp[threadIdx.x] = op; si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int k = *npicks;
        if (si[0] >= 0) { picks[k] = si[0]; alive[si[0]] = 0; *npicks = k + 1; picks[max_picks] = si[0]; }
        else picks[max_picks] = -1;                           // nothing left: later suppress passes are no-ops
    }
}

__global__ void k_nms_suppress(const double* __restrict__ boxes, int* __restrict__ alive, int n,
                               const int* __restrict__ picks, int max_picks, double thresh, double ax, double ay) {
    const int cur = picks[max_picks];
    if (cur < 0) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n || !alive[j]) return;
    const double* b = boxes + (size_t)j * 7;
    if (b[0] - ax < 0 || b[0] + ax > 100 || b[1] - ay < 0 || b[1] + ay > 100) { alive[j] = 0; return; }
    if (rot_iou(boxes + (size_t)cur * 7, b) > thresh) alive[j] = 0;
}

__global__ void k_nms_gather(const double* __restrict__ boxes, const double* __restrict__ probs,
                             const int* __restrict__ picks, const int* __restrict__ npicks,
                             double* __restrict__ out_boxes, double* __restrict__ out_probs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= *npicks) return;
    for (int c = 0; c < 7; ++c) out_boxes[k * 7 + c] = boxes[(size_t)picks[k] * 7 + c];
    out_probs[k] = probs[picks[k]];
}

// ---- label generation (serialize_data.py:194-307) -----------------------------------------------------------
struct Anchor {
    bool in_range;
    double box[7];
    int xV, yV, a, order;
};

__device__ __forceinline__ Anchor make_anchor(const lisec_rpn_cfg& cfg, int i) {
    Anchor A;
    const int M = cfg.outX * cfg.outY;
    A.a = i / M;
    const int m = i - A.a * M;
    A.xV = m / cfg.outY - cfg.outX / 2;                       // range(int(-outX/2), int(outX/2)) (:223)
    A.yV = m % cfg.outY - cfg.outY / 2;
    A.order = i;                                              // loop nest order: anchor, xVoxel, yVoxel
    const double* an = cfg.anchors[A.a];
    const double cX = cfg.vx * A.xV + cfg.vx / 2, cY = cfg.vy * A.yV + cfg.vy / 2;
    A.in_range = !(cX - an[0] / 2 < cfg.vx * (-(cfg.outX / 2)) || cX + an[0] / 2 > cfg.vx * (cfg.outX / 2) ||
                   cY - an[1] / 2 < cfg.vy * (-(cfg.outY / 2)) || cY + an[1] / 2 > cfg.vy * (cfg.outY / 2));
    A.box[0] = cX; A.box[1] = cY; A.box[2] = 1.0;             // centerZ hard set to 1 (:221)
    A.box[3] = an[0]; A.box[4] = an[1]; A.box[5] = an[2]; A.box[6] = an[3];
    return A;
}

__device__ __forceinline__ void regression(const double* ab, const double* fb, double* t) {
    t[0] = (fb[0] - ab[0]) / ab[3]; t[1] = (fb[1] - ab[1]) / ab[4]; t[2] = (fb[2] - ab[2]) / ab[5];
    t[3] = log(fb[3] / ab[3]); t[4] = log(fb[4] / ab[4]); t[5] = log(fb[5] / ab[5]);
    t[6] = fb[6] - ab[6];
}

__device__ __forceinline__ size_t wrapped_index(const lisec_rpn_cfg& cfg, int xV, int yV) {
    // the reference indexes numpy arrays with NEGATIVE xVoxel / yVoxel: they wrap (:284-294)
    const int ix = xV < 0 ? xV + cfg.outX : xV, iy = yV < 0 ? yV + cfg.outY : yV;
    return (size_t)ix * cfg.outY + iy;
}

// one thread per anchor, boxes visited in order (the per-anchor state machine of :235-294)
__global__ void k_label_anchors(lisec_rpn_cfg cfg, const double* __restrict__ fixed, int B, double iou_lo,
                                double iou_hi, double* __restrict__ valid, double* __restrict__ overlap,
                                double* __restrict__ out_reg, unsigned long long* __restrict__ best_iou_bits,
                                int* __restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * cfg.outX * cfg.outY) return;
    const Anchor A = make_anchor(cfg, i);
    if (!A.in_range) return;
    int type = 0;                                             // 0 neg, 1 neutral, 2 pos
    double best_loc = 0.0, best_reg[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
        const double* fb = fixed + (size_t)b * 7;
        const double iou = rot_iou(A.box, fb);
        if (iou > 0.0) atomicMax(&best_iou_bits[b], (unsigned long long)__double_as_longlong(iou));
        if (iou >= iou_hi) {
            type = 2;
            atomicAdd(&count[b], 1);
            if (iou > best_loc) { best_loc = iou; regression(A.box, fb, best_reg); }
        }
        if (iou_lo < iou && iou <= iou_hi && type != 2) type = 1;
    }
    const size_t at = wrapped_index(cfg, A.xV, A.yV) * 2 + A.a;
    if (type == 0) { valid[at] = 1.0; overlap[at] = 0.0; }
    else if (type == 1) { valid[at] = 0.0; overlap[at] = 0.0; }
    else {
        valid[at] = 1.0; overlap[at] = 1.0;
        double* o = out_reg + wrapped_index(cfg, A.xV, A.yV) * 14 + A.a * 7;
        for (int c = 0; c < 7; ++c) o[c] = best_reg[c];
    }
}

// first anchor (in loop order) attaining each box's best IoU: `iou > bestIouForBox` keeps the first (:266-269)
__global__ void k_label_best_order(lisec_rpn_cfg cfg, const double* __restrict__ fixed, int B,
                                   const unsigned long long* __restrict__ best_iou_bits,
                                   int* __restrict__ best_order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * cfg.outX * cfg.outY) return;
    const Anchor A = make_anchor(cfg, i);
    if (!A.in_range) return;
    for (int b = 0; b < B; ++b) {
        const unsigned long long best = best_iou_bits[b];
        if (best == 0ull) continue;
        const double iou = rot_iou(A.box, fixed + (size_t)b * 7);
        if (iou > 0.0 && (unsigned long long)__double_as_longlong(iou) == best) atomicMin(&best_order[b], A.order);
    }
}

// boxes without a positive anchor get their best one (:297-307), in box order like the reference loop
__global__ void k_label_fixup(lisec_rpn_cfg cfg, const double* __restrict__ fixed, int B,
                              const unsigned long long* __restrict__ best_iou_bits,
                              const int* __restrict__ best_order, const int* __restrict__ count,
                              double* __restrict__ valid, double* __restrict__ overlap, double* __restrict__ out_reg) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int b = 0; b < B; ++b) {
        if (count[b] != 0 || best_iou_bits[b] == 0ull) continue;
        const Anchor A = make_anchor(cfg, best_order[b]);
        double t[7];
        regression(A.box, fixed + (size_t)b * 7, t);
        const size_t cell = wrapped_index(cfg, A.xV, A.yV);
        valid[cell * 2 + A.a] = 1.0;
        overlap[cell * 2 + A.a] = 1.0;
        for (int c = 0; c < 7; ++c) out_reg[cell * 14 + A.a * 7 + c] = t[c];
    }
}

struct BoxWs {
    double *boxes, *probs;
    int *alive, *picks, *npicks;
    size_t bytes;
    BoxWs(void* base, int n, int max_picks) {
        Carver c(base);
        boxes = c.take<double>((size_t)n * 7);
        probs = c.take<double>(n);
        alive = c.take<int>(n);
        picks = c.take<int>(max_picks + 1);
        npicks = c.take<int>(4);
        bytes = c.off;
    }
};

int check_cfg(const lisec_rpn_cfg* cfg) {
    LISEC_CHECK_ARG(cfg && cfg->outX > 0 && cfg->outY > 0 && cfg->outX % 2 == 0 && cfg->outY % 2 == 0 &&
                    cfg->vx > 0 && cfg->vy > 0, "bad RPN grid configuration");
    return 0;
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" size_t lisec_rpn_to_region_workspace_bytes(const lisec_rpn_cfg* cfg, int max_boxes) {
    if (check_cfg(cfg) || max_boxes < 0) return 0;
    return BoxWs(nullptr, 2 * cfg->outX * cfg->outY, max_boxes + 1).bytes;
}

extern "C" int lisec_rpn_to_region(const lisec_rpn_cfg* cfg, const float* cls, int cls_stride, const float* reg,
                                   int reg_stride, double overlap_thresh, int max_boxes, void* workspace,
                                   size_t workspace_bytes, double* out_boxes, double* out_probs, int32_t* out_count,
                                   lisec_stream_t stream_) {
    if (int rc = check_cfg(cfg)) return rc;
    LISEC_CHECK_ARG(cls && reg && workspace && out_boxes && out_probs && out_count && max_boxes >= 0 &&
                    cls_stride >= 2 && reg_stride >= 14, "bad arguments");
    const int n = 2 * cfg->outX * cfg->outY, max_picks = max_boxes + 1;   // the loop breaks once len(pick) > maxBoxes
    BoxWs ws(workspace, n, max_picks);
    if (workspace_bytes < ws.bytes) {
        set_error("rpn_to_region workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    LISEC_HIP_TRY(hipMemsetAsync(ws.npicks, 0, sizeof(int) * 4, st));
    LISEC_LAUNCH(k_rpn_decode, dim3(cdiv(n, 256)), dim3(256), 0, st, cls, cls_stride, reg, reg_stride, *cfg,
                       ws.boxes, ws.probs, ws.alive);
    for (int k = 0; k < max_picks; ++k) {
        LISEC_LAUNCH(k_nms_pick, dim3(1), dim3(1024), 0, st, ws.probs, ws.alive, n, ws.picks, ws.npicks, max_picks);
        LISEC_LAUNCH(k_nms_suppress, dim3(cdiv(n, 256)), dim3(256), 0, st, ws.boxes, ws.alive, n, ws.picks,
                           max_picks, overlap_thresh, cfg->anchors[0][0], cfg->anchors[0][1]);
    }
    LISEC_LAUNCH(k_nms_gather, dim3(1), dim3(256), 0, st, ws.boxes, ws.probs, ws.picks, ws.npicks, out_boxes,
                       out_probs);
    LISEC_HIP_TRY(hipMemcpyAsync(out_count, ws.npicks, sizeof(int), hipMemcpyDeviceToDevice, st));
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_rpn_decode(const lisec_rpn_cfg* cfg, const float* cls, int cls_stride, const float* reg, int reg_stride,
                                double* boxes, double* probs, int32_t* legal, lisec_stream_t stream_) {
    if (int rc = check_cfg(cfg)) return rc;
    LISEC_CHECK_ARG(cls && reg && boxes && probs && legal && cls_stride >= 2 && reg_stride >= 14, "bad arguments");
    const int n = 2 * cfg->outX * cfg->outY;
    LISEC_LAUNCH(k_rpn_decode, dim3(cdiv(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), cls, cls_stride, reg,
                 reg_stride, *cfg, boxes, probs, legal);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" int lisec_box_geometry(const double* boxes, int n, const double* pair_area, double* corners, double* pair_out,
                                  lisec_stream_t stream_) {
    LISEC_CHECK_ARG(boxes && corners && pair_out && n >= 1, "bad arguments");
    LISEC_LAUNCH(k_box_geometry, dim3(cdiv(n, 64)), dim3(64), 0, static_cast<hipStream_t>(stream_), boxes, n, pair_area,
                 corners, pair_out);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}

extern "C" size_t lisec_rpn_labels_workspace_bytes(int n_boxes) {
    if (n_boxes < 0) return 0;
    return align_up(sizeof(unsigned long long) * (size_t)(n_boxes + 1), 256) + 2 * align_up(sizeof(int) * (size_t)(n_boxes + 1), 256);
}

extern "C" int lisec_rpn_labels(const lisec_rpn_cfg* cfg, const double* fixed_boxes, int n_boxes, double iou_lo,
                                double iou_hi, void* workspace, size_t workspace_bytes, double* valid,
                                double* overlap, double* out_regress, lisec_stream_t stream_) {
    if (int rc = check_cfg(cfg)) return rc;
    LISEC_CHECK_ARG(n_boxes >= 0 && workspace && valid && overlap && out_regress && (n_boxes == 0 || fixed_boxes),
                    "bad arguments");
    if (workspace_bytes < lisec_rpn_labels_workspace_bytes(n_boxes)) {
        set_error("rpn_labels workspace too small");
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    Carver c(workspace);
    unsigned long long* best_bits = c.take<unsigned long long>(n_boxes + 1);
    int* best_order = c.take<int>(n_boxes + 1);
    int* count = c.take<int>(n_boxes + 1);
    const size_t cells = (size_t)cfg->outX * cfg->outY;
    LISEC_HIP_TRY(hipMemsetAsync(valid, 0, sizeof(double) * cells * 2, st));
    LISEC_HIP_TRY(hipMemsetAsync(overlap, 0, sizeof(double) * cells * 2, st));
    LISEC_HIP_TRY(hipMemsetAsync(out_regress, 0, sizeof(double) * cells * 14, st));
    LISEC_HIP_TRY(hipMemsetAsync(best_bits, 0, sizeof(unsigned long long) * (n_boxes + 1), st));
    LISEC_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int) * (n_boxes + 1), st));
    LISEC_HIP_TRY(hipMemsetAsync(best_order, 0x7f, sizeof(int) * (n_boxes + 1), st));
    const int n = 2 * (int)cells;
    LISEC_LAUNCH(k_label_anchors, dim3(cdiv(n, 128)), dim3(128), 0, st, *cfg, fixed_boxes, n_boxes, iou_lo, iou_hi,
                       valid, overlap, out_regress, best_bits, count);
    if (n_boxes > 0) {
        LISEC_LAUNCH(k_label_best_order, dim3(cdiv(n, 128)), dim3(128), 0, st, *cfg, fixed_boxes, n_boxes, best_bits,
                           best_order);
        LISEC_LAUNCH(k_label_fixup, dim3(1), dim3(64), 0, st, *cfg, fixed_boxes, n_boxes, best_bits, best_order,
                           count, valid, overlap, out_regress);
    }
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
