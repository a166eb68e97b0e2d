// Lidar ingest for gfx950: per-sensor rotate + translate of the raw .bin points, written straight into the
// float64 (n,3) cloud the voxeliser reads (reference combine_lidar_data / rotate_points,
// model_training.py:65-98: rawPoints.reshape(-1,5)[:, :3] -> np.dot(R, p.T).T + translation, float64).
// HBM-bound: 20 B read + 24 B written per point.
#include "common.h"

namespace lisec {
namespace {

struct Pose { double r[9]; double t[3]; };

__global__ void k_lidar_transform(const float* __restrict__ raw, int n, int raw_stride, Pose ps,
                                  double* __restrict__ out) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float* p = raw + i * raw_stride;
        const double x = (double)p[0], y = (double)p[1], z = (double)p[2];
        // row-by-row dot product in float64, then the sensor translation (model_training.py:93-94)
        double* o = out + i * 3;
        o[0] = (ps.r[0] * x + ps.r[1] * y + ps.r[2] * z) + ps.t[0];
        o[1] = (ps.r[3] * x + ps.r[4] * y + ps.r[5] * z) + ps.t[1];
        o[2] = (ps.r[6] * x + ps.r[7] * y + ps.r[8] * z) + ps.t[2];
    }
}

}  // namespace
}  // namespace lisec

using namespace lisec;

extern "C" int lisec_lidar_transform(const float* raw, int n_points, int raw_stride, const double* rotation9,
                                     const double* translation3, double* out, lisec_stream_t stream_) {
    LISEC_CHECK_ARG(n_points >= 0 && raw_stride >= 3 && rotation9 && translation3, "bad arguments");
    if (n_points == 0) return LISEC_OK;
    LISEC_CHECK_ARG(raw && out, "NULL pointer");
    Pose ps;
    for (int i = 0; i < 9; ++i) ps.r[i] = rotation9[i];
    for (int i = 0; i < 3; ++i) ps.t[i] = translation3[i];
    int gb = cdiv(n_points, 256);
    if (gb > 2048) gb = 2048;
    LISEC_LAUNCH(k_lidar_transform, dim3(gb), dim3(256), 0, static_cast<hipStream_t>(stream_), raw, n_points,
                       raw_stride, ps, out);
    LISEC_LAUNCH_CHECK();
    return LISEC_OK;
}
