// Voxeliser for gfx950: points -> occupied voxels + compact per-voxel feature rows.
//
// Replaces get_voxel / VFE_preprocessing / sparse.to_dense of the reference
// (model_training.py:103-152, :279).  The reference buckets points in a Python dict and
// then materialises a (8,200,400,35,6) dense tensor; here the cloud is bucketed with a
// counting sort over the grid cells and only real rows are written (24 B per kept point).
//
// Pipeline (all on one stream, no host sync):
//   memset cell_count                       2.56 MB for the Lyft grid
//   k_key_count      1 thread / point       key = floor(p/size) in fp64, strict range test
//                                           (model_training.py:117-122), atomic arrival rank
//   k_cell_totals  \
//   k_scan_totals   > exclusive scan over the NZ*NX*NY cells: voxel ordinal, point offset,
//   k_cell_assign  /  row offset -> voxels come out sorted by cell = (z*NX + x)*NY + y
//   k_place          1 thread / point       bucket[pt_start[v] + rank] = point index
//   k_features       1 wave / voxel         keeps the `sampleSize` lowest point indices in
//                                           ascending order (deterministic stand-in for the
//                                           unseeded np.random.choice, model_training.py:132),
//                                           centroid = sequential fp64 sum / s (np.mean order,
//                                           :135), rows [x,y,z,x-cx,y-cy,z-cz] rounded to fp32;
//                                           on the side the 6 + 21 first / second moments of the
//                                           rows it writes (row_stats: what the VFE's first
//                                           BatchNormalization needs, see lisec_hip.h)
// All of it is HBM/latency-bound integer work: no MFMA, LDS only as per-wave scratch.
#include "common.h"

namespace lisec {
namespace {

constexpr int kScanThreads = 256;
constexpr int kCellsPerThread = 8;
constexpr int kCellsPerBlock = kScanThreads * kCellsPerThread;

struct GridDims {
    int nx, ny, nz, ncells, T;
    double xs, ys, zs;
    int mx, my, mz;
};

// cell_count[ncells] = 0 and info[8] = 0 in ONE launch (two hipMemsetAsync calls are two fill kernels: ~11 us + a gap)
__global__ void __launch_bounds__(256)
k_zero_counts(int* __restrict__ cell_count, int ncells, int* __restrict__ info) {
    const int n4 = ncells >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256)
        reinterpret_cast<int4*>(cell_count)[i] = make_int4(0, 0, 0, 0);
    if (blockIdx.x == 0) {
        if (threadIdx.x < 8) info[threadIdx.x] = 0;
        if ((int)threadIdx.x < (ncells & 3)) cell_count[(n4 << 2) + threadIdx.x] = 0;
    }
}

template <typename T>
__global__ void k_key_count(const T* __restrict__ pts, int n, int stride, GridDims g,
                            int* __restrict__ cell_count, int* __restrict__ key,
                            int* __restrict__ rank) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T* p = pts + (size_t)i * stride;
    // fp64 divide + floor exactly as get_voxel does on float64 arrays (model_training.py:104-106)
    double kx = floor((double)p[0] / g.xs);
    double ky = floor((double)p[1] / g.ys);
    double kz = floor((double)p[2] / g.zs);
    bool ok = (-(double)g.mx < kx) && (kx < (double)g.mx) && (-(double)g.my < ky) &&
              (ky < (double)g.my) && (0.0 < kz) && (kz < (double)g.mz);   // NaN -> false
    int lin = -1, r = 0;
    if (ok) {
        int fx = (int)kx + g.mx, fy = (int)ky + g.my, fz = (int)kz;        // "fixedKey" (:122)
        lin = (fz * g.nx + fx) * g.ny + fy;
        r = atomicAdd(&cell_count[lin], 1);
    }
    key[i] = lin;
    rank[i] = r;
}

// per block: (#occupied cells, #points, #kept rows)
__global__ void k_cell_totals(const int* __restrict__ cell_count, GridDims g,
                              int* __restrict__ totals) {
    __shared__ int red[3][kScanThreads / 64];
    int base = blockIdx.x * kCellsPerBlock + threadIdx.x * kCellsPerThread;
    int occ = 0, cnt = 0, rows = 0;
#pragma unroll
    for (int j = 0; j < kCellsPerThread; ++j) {
        int c = base + j;
        int v = c < g.ncells ? cell_count[c] : 0;
        occ += v > 0;
        cnt += v;
        rows += v < g.T ? v : g.T;
    }
    occ = wave_sum(occ); cnt = wave_sum(cnt); rows = wave_sum(rows);
    int w = threadIdx.x >> 6;
    if (lane_id() == 0) { red[0][w] = occ; red[1][w] = cnt; red[2][w] = rows; }
    __syncthreads();
    if (threadIdx.x < 3) {
        int s = 0;
        for (int k = 0; k < kScanThreads / 64; ++k) s += red[threadIdx.x][k];
        totals[blockIdx.x * 3 + threadIdx.x] = s;
    }
}

// single block: exclusive scan of the per-block totals, header words
__global__ void k_scan_totals(int* __restrict__ totals, int nblk, int cap_voxels,
                              int* __restrict__ info, int* __restrict__ row_start,
                              long long* __restrict__ row_stats) {
    __shared__ int carry[3];
    __shared__ int wsum[3][16];
    if (row_stats)                                   // moments + the VFE's scratch start every sweep at zero
        for (int i = threadIdx.x; i < LISEC_ROW_STATS_WORDS; i += blockDim.x) row_stats[i] = 0;
    if (threadIdx.x < 3) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += blockDim.x) {
        int b = b0 + threadIdx.x;
        int v[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) v[q] = b < nblk ? totals[b * 3 + q] : 0;
        int inc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int x = v[q];
            for (int o = 1; o < 64; o <<= 1) {          // inclusive scan inside the wave
                int y = __shfl_up(x, o, 64);
                if (lane_id() >= o) x += y;
            }
            inc[q] = x;
            if (lane_id() == 63) wsum[q][threadIdx.x >> 6] = x;
        }
        __syncthreads();
        int nw = blockDim.x >> 6, w = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int pre = carry[q];
            for (int k = 0; k < w; ++k) pre += wsum[q][k];
            if (b < nblk) totals[b * 3 + q] = pre + inc[q] - v[q];   // exclusive
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            int s = carry[threadIdx.x];
            for (int k = 0; k < nw; ++k) s += wsum[threadIdx.x][k];
            carry[threadIdx.x] = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int V = carry[0];
        info[LISEC_VI_NVOX] = V;
        info[LISEC_VI_NVALID] = carry[1];
        info[LISEC_VI_NROWS] = carry[2];
        if (V > cap_voxels) info[LISEC_VI_OVERFLOW] = 1;
        else row_start[V] = carry[2];
    }
}

__global__ void k_cell_assign(const int* __restrict__ cell_count, const int* __restrict__ totals,
                              GridDims g, int cap_voxels, int* __restrict__ cell_voxel,
                              int* __restrict__ coords, int* __restrict__ counts,
                              int* __restrict__ npts, int* __restrict__ pt_start,
                              int* __restrict__ row_start, int* __restrict__ info) {
    __shared__ int wsum[3][kScanThreads / 64];
    int base = blockIdx.x * kCellsPerBlock + threadIdx.x * kCellsPerThread;
    int cv[kCellsPerThread];
    int t[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < kCellsPerThread; ++j) {
        int c = base + j;
        cv[j] = c < g.ncells ? cell_count[c] : 0;
        t[0] += cv[j] > 0;
        t[1] += cv[j];
        t[2] += cv[j] < g.T ? cv[j] : g.T;
    }
    int ex[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int x = t[q];
        for (int o = 1; o < 64; o <<= 1) {
            int y = __shfl_up(x, o, 64);
            if (lane_id() >= o) x += y;
        }
        if (lane_id() == 63) wsum[q][threadIdx.x >> 6] = x;
        ex[q] = x - t[q];
    }
    __syncthreads();
    int w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int pre = totals[blockIdx.x * 3 + q];
        for (int k = 0; k < w; ++k) pre += wsum[q][k];
        ex[q] += pre;
    }
    int mx = 0;
#pragma unroll
    for (int j = 0; j < kCellsPerThread; ++j) {
        int c = base + j;
        if (c >= g.ncells) break;
        int n = cv[j];
        if (n > 0) {
            int v = ex[0];
            if (v < cap_voxels) {
                cell_voxel[c] = v;
                int y = c % g.ny, x = (c / g.ny) % g.nx, z = c / (g.ny * g.nx);
                coords[v * 3 + 0] = z;          // (z, x, y): model_training.py:148
                coords[v * 3 + 1] = x;
                coords[v * 3 + 2] = y;
                counts[v] = n;
                npts[v] = n < g.T ? n : g.T;
                pt_start[v] = ex[1];
                row_start[v] = ex[2];
            } else {
                cell_voxel[c] = -1;
            }
            ex[0] += 1;
            ex[1] += n;
            ex[2] += n < g.T ? n : g.T;
            mx = n > mx ? n : mx;
        } else {
            cell_voxel[c] = -1;
        }
    }
    mx = wave_max(mx);
    if (lane_id() == 0 && mx > 0) atomicMax(&info[LISEC_VI_MAXCOUNT], mx);
}

__global__ void k_place(const int* __restrict__ key, const int* __restrict__ rank, int n,
                        const int* __restrict__ cell_voxel, const int* __restrict__ pt_start,
                        int* __restrict__ bucket) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int k = key[i];
    if (k < 0) return;
    int v = cell_voxel[k];
    if (v < 0) return;                      // only on capacity overflow
    bucket[pt_start[v] + rank[i]] = i;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_features(const T* __restrict__ pts, int stride, const int* __restrict__ info, int cap_voxels,
           int Tmax, const int* __restrict__ counts, const int* __restrict__ pt_start,
           const int* __restrict__ row_start, const int* __restrict__ bucket,
           float* __restrict__ rows, int* __restrict__ row_point, long long* __restrict__ row_stats) {
    __shared__ double spt[4][64][3];
    __shared__ int ssel[4][64];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    // row moments (row_stats): lane k < 27 owns ONE moment -- sum x_k for k < 6, else sum x_a*x_b for the k-th pair
    // (a <= b) -- and walks the rows of each voxel through LDS, so no cross-lane reduction is ever needed
    __shared__ float sfeat[4][64][6];
    double mom = 0.0;
    int ma = lane < 6 ? lane : 0, mb = -1;
    if (lane >= 6 && lane < 27) {
        int q = lane - 6;
        ma = 0;
        while (q >= 6 - ma) { q -= 6 - ma; ++ma; }      // pairs in the order (0,0) (0,1) .. (0,5) (1,1) .. (5,5)
        mb = ma + q;
    }
    int V = info[LISEC_VI_NVOX];
    if (V > cap_voxels) V = cap_voxels;
    const int nwaves = gridDim.x * 4;
    for (int v = blockIdx.x * 4 + w; v < V; v += nwaves) {
        const int c = counts[v], b0 = pt_start[v];
        const int s = c < Tmax ? c : Tmax;
        int mine = -1;                       // lane t < s ends up with the t-th smallest index
        if (c <= 64) {
            int my = lane < c ? bucket[b0 + lane] : 0x7fffffff;
            int r = 0;
            for (int j = 0; j < c; ++j) r += __shfl(my, j, 64) < my;
            if (lane < c && r < s) ssel[w][r] = my;
            __threadfence_block();
            if (lane < s) mine = ssel[w][lane];
        } else {
            int prev = -1;
            for (int k = 0; k < s; ++k) {
                int m = 0x7fffffff;
                for (int j = lane; j < c; j += 64) {
                    int b = bucket[b0 + j];
                    if (b > prev && b < m) m = b;
                }
                m = wave_min(m);
                if (lane == k) mine = m;
                prev = m;
            }
        }
        double px = 0, py = 0, pz = 0;
        if (lane < s) {
            const T* p = pts + (size_t)mine * stride;
            px = (double)p[0]; py = (double)p[1]; pz = (double)p[2];
            spt[w][lane][0] = px; spt[w][lane][1] = py; spt[w][lane][2] = pz;
        }
        __threadfence_block();
        // centroid: rows added one after the other, then one divide (np.mean, :135)
        double acc = 0.0;
        if (lane < 3) {
            for (int r = 0; r < s; ++r) acc = acc + spt[w][r][lane];
            acc = acc / (double)s;
        }
        double cx = __shfl(acc, 0, 64), cy = __shfl(acc, 1, 64), cz = __shfl(acc, 2, 64);
        if (lane < s) {
            int row = row_start[v] + lane;
            float* o = rows + (size_t)row * 6;
            const float fr[6] = {(float)px, (float)py, (float)pz, (float)(px - cx), (float)(py - cy), (float)(pz - cz)};
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = fr[k];
            if (row_point) row_point[row] = mine;
            if (row_stats) {                 // the fp32 values the VFE will read
#pragma unroll
                for (int k = 0; k < 6; ++k) sfeat[w][lane][k] = fr[k];
            }
        }
        if (row_stats) {
            __threadfence_block();
            if (lane < 27) {
                for (int r = 0; r < s; ++r)
                    mom += (double)sfeat[w][r][ma] * (mb < 0 ? 1.0 : (double)sfeat[w][r][mb]);
            }
        }
        __threadfence_block();               // LDS scratch is reused by the next voxel
    }
    if (row_stats) {
        __shared__ double smom[4][27];
        if (lane < 27) smom[w][lane] = mom;
        __syncthreads();
        if (threadIdx.x < 27) {
            const double t = ((smom[0][threadIdx.x] + smom[1][threadIdx.x]) + smom[2][threadIdx.x]) + smom[3][threadIdx.x];
            if (t != 0.0)
                fx_atomic_add(row_stats + ((size_t)(blockIdx.x % LISEC_ROW_STATS_REPLICAS) * 27 + threadIdx.x) * 2, t);
        }
    }
}

__global__ void k_rows_to_padded(const int* __restrict__ info, int cap_voxels,
                                 const int* __restrict__ npts, const int* __restrict__ row_start,
                                 const float* __restrict__ rows, int T, float* __restrict__ padded) {
    int V = info[LISEC_VI_NVOX];
    if (V > cap_voxels) V = cap_voxels;
    long long total = (long long)V * T * 6;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int f = (int)(i % 6);
        int t = (int)((i / 6) % T);
        int v = (int)(i / (6LL * T));
        padded[i] = t < npts[v] ? rows[((size_t)row_start[v] + t) * 6 + f] : 0.0f;
    }
}

int make_dims(const lisec_voxel_cfg* cfg, GridDims* g) {
    LISEC_CHECK_ARG(cfg, "cfg is NULL");
    LISEC_CHECK_ARG(cfg->xSize > 0 && cfg->ySize > 0 && cfg->zSize > 0, "voxel sizes must be > 0");
    LISEC_CHECK_ARG(cfg->maxVoxelX > 0 && cfg->maxVoxelY > 0 && cfg->maxVoxelZ > 0, "bad grid extent");
    LISEC_CHECK_ARG(cfg->sampleSize >= 1 && cfg->sampleSize <= 64, "sampleSize must be in [1,64]");
    long long nc = 4LL * cfg->maxVoxelX * cfg->maxVoxelY * cfg->maxVoxelZ;
    LISEC_CHECK_ARG(nc < (1LL << 30), "grid too large");
    g->nx = 2 * cfg->maxVoxelX; g->ny = 2 * cfg->maxVoxelY; g->nz = cfg->maxVoxelZ;
    g->ncells = (int)nc; g->T = cfg->sampleSize;
    g->xs = cfg->xSize; g->ys = cfg->ySize; g->zs = cfg->zSize;
    g->mx = cfg->maxVoxelX; g->my = cfg->maxVoxelY; g->mz = cfg->maxVoxelZ;
    return 0;
}

struct VoxWs {
    int *cell_count, *totals, *key, *rank, *bucket, *pt_start;
    size_t bytes;
};

VoxWs carve(void* ws, const GridDims& g, int n, int cap_voxels) {
    Carver c(ws);
    VoxWs w;
    int nblk = cdiv(g.ncells, kCellsPerBlock);
    w.cell_count = c.take<int>(g.ncells);
    w.totals = c.take<int>((size_t)nblk * 3);
    w.key = c.take<int>(n > 0 ? n : 1);
    w.rank = c.take<int>(n > 0 ? n : 1);
    w.bucket = c.take<int>(n > 0 ? n : 1);
    w.pt_start = c.take<int>(cap_voxels > 0 ? cap_voxels : 1);
    w.bytes = c.off;
    return w;
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" size_t lisec_voxelize_workspace_bytes(const lisec_voxel_cfg* cfg, int n_points) {
    GridDims g;
    if (make_dims(cfg, &g) != 0 || n_points < 0) return 0;
    // pt_start is sized for the worst case cap_voxels = n_points
    return carve(nullptr, g, n_points, n_points).bytes;
}

extern "C" int lisec_voxelize(const lisec_voxel_cfg* cfg, const void* points, int dtype,
                              int n_points, int point_stride, void* workspace,
                              size_t workspace_bytes, int cap_voxels, int32_t* info,
                              int32_t* cell_voxel, int32_t* coords, int32_t* counts,
                              int32_t* npts, int32_t* row_start, float* rows, int32_t* row_point,
                              int64_t* row_stats, lisec_stream_t stream_) {
    GridDims g;
    if (int rc = make_dims(cfg, &g)) return rc;
    LISEC_CHECK_ARG(n_points >= 0 && point_stride >= 3, "n_points/point_stride invalid");
    LISEC_CHECK_ARG(dtype == 0 || dtype == 1, "dtype must be 0 (f32) or 1 (f64)");
    LISEC_CHECK_ARG(cap_voxels >= 0 && cap_voxels <= n_points, "cap_voxels must be in [0, n_points]");
    LISEC_CHECK_ARG(info && cell_voxel && coords && counts && npts && row_start && rows && workspace,
                    "NULL output/workspace pointer");
    LISEC_CHECK_ARG(n_points == 0 || points, "points is NULL");
    VoxWs w = carve(workspace, g, n_points, cap_voxels);
    if (workspace_bytes < w.bytes) {
        set_error("voxelize workspace too small: %zu < %zu", workspace_bytes, w.bytes);
        return LISEC_ENOSPC;
    }
    hipStream_t st = static_cast<hipStream_t>(stream_);
    {
        int zb = cdiv(g.ncells >> 2, 256);
        if (zb > 1024) zb = 1024;
        LISEC_LAUNCH(k_zero_counts, dim3(zb < 1 ? 1 : zb), dim3(256), 0, st, w.cell_count, g.ncells, info);
    }
    const int nblk = cdiv(g.ncells, kCellsPerBlock);
    if (n_points > 0) {
        int gb = cdiv(n_points, 256);
        if (dtype == 0)
            LISEC_LAUNCH(k_key_count<float>, dim3(gb), dim3(256), 0, st, (const float*)points,
                               n_points, point_stride, g, w.cell_count, w.key, w.rank);
        else
            LISEC_LAUNCH(k_key_count<double>, dim3(gb), dim3(256), 0, st, (const double*)points,
                               n_points, point_stride, g, w.cell_count, w.key, w.rank);
        LISEC_LAUNCH_CHECK();
    }
    LISEC_LAUNCH(k_cell_totals, dim3(nblk), dim3(kScanThreads), 0, st, w.cell_count, g, w.totals);
    long long* stats = reinterpret_cast<long long*>(row_stats);
    LISEC_LAUNCH(k_scan_totals, dim3(1), dim3(1024), 0, st, w.totals, nblk, cap_voxels, info,
                       row_start, stats);
    LISEC_LAUNCH(k_cell_assign, dim3(nblk), dim3(kScanThreads), 0, st, w.cell_count, w.totals, g,
                       cap_voxels, cell_voxel, coords, counts, npts, w.pt_start, row_start, info);
    LISEC_LAUNCH_CHECK();
    if (n_points > 0 && cap_voxels > 0) {
        int gb = cdiv(n_points, 256);
        LISEC_LAUNCH(k_place, dim3(gb), dim3(256), 0, st, w.key, w.rank, n_points, cell_voxel,
                           w.pt_start, w.bucket);
        int fb = cdiv(cap_voxels, 4);
        if (fb > 2048) fb = 2048;
        if (dtype == 0)
            LISEC_LAUNCH(k_features<float>, dim3(fb), dim3(256), 0, st, (const float*)points,
                               point_stride, info, cap_voxels, g.T, counts, w.pt_start, row_start,
                               w.bucket, rows, row_point, stats);
        else
            LISEC_LAUNCH(k_features<double>, dim3(fb), dim3(256), 0, st, (const double*)points,
                               point_stride, info, cap_voxels, g.T, counts, w.pt_start, row_start,
                               w.bucket, rows, row_point, stats);
        LISEC_LAUNCH_CHECK();
    }
    return LISEC_OK;
}

extern "C" int lisec_voxel_rows_to_padded(const int32_t* info, const int32_t* npts,
                                          const int32_t* row_start, const float* rows, int sampleSize,
                                          int cap_voxels, float* padded, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(info && npts && row_start && rows && padded, "NULL pointer");
    LISEC_CHECK_ARG(sampleSize >= 1 && sampleSize <= 64 && cap_voxels >= 0, "bad sizes");
    if (cap_voxels == 0) return LISEC_OK;
    hipStream_t st = static_cast<hipStream_t>(stream_);
    long long total = (long long)cap_voxels * sampleSize * 6;
    int gb = cdiv(total, 256);
    if (gb > 4096) gb = 4096;
    LISEC_LAUNCH(k_rows_to_padded, dim3(gb), dim3(256), 0, st, info, cap_voxels, npts, row_start,
                       rows, sampleSize, padded);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
