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
};

int conv_geom_check(const lisec_conv_geom* c, ConvGeom* g);

}  // namespace lisec
