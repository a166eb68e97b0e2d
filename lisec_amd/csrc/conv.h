// Internal: geometry shared by the implicit-GEMM kernels (igemm.hip, wgrad.hip).
#pragma once
#include "common.h"

namespace lisec {

struct ConvGeom {
    int Di, Hi, Wi;          // tensor that is gathered from
    int Do, Ho, Wo;          // tensor that is written (M = Do*Ho*Wo rows)
    int KD, KH, KW;
    int ls_d, ls_h, ls_w;    // log2(stride)
    int pd, ph, pw;
    int Cin, in_stride, Cout, out_stride, CoutP;
    int M;
    int ps, ps_channels;
};

int conv_geom_check(const lisec_conv_geom* c, ConvGeom* g);

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

// XCD-aware bijective tile remap: tiles that are neighbours in memory stay on one XCD's L2
__device__ __forceinline__ int xcd_remap(int orig, int n) {
    const int q = n >> 3, r = n & 7, xcd = orig & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
#endif

}  // namespace lisec
