"""Weight-gradient kernels alone on the Lyft-grid layer shapes (GPU box only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import run_wgrad

if __name__ == "__main__":
    run_wgrad("mid2 wgrad 64->64 s1", 0, (4, 200, 400), (2, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64, iters=20)
    run_wgrad("mid3 wgrad 64->64 s(2,1,1)", 0, (2, 200, 400), (1, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, iters=20)
    run_wgrad("rpn1 wgrad 128->128", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, iters=20, in_bn=True)
    run_wgrad("rpn2 wgrad 128->128", 0, (1, 50, 100), (1, 50, 100), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, iters=20, in_bn=True)
    run_wgrad("rpn3 wgrad 256->256", 0, (1, 25, 50), (1, 25, 50), (1, 3, 3), (1, 1, 1), (0, 1, 1), 256, 256, iters=20, in_bn=True)
    run_wgrad("rpn1.conv0 wgrad s2 64->128", 0, (1, 200, 400), (1, 100, 200), (1, 3, 3), (1, 2, 2), (0, 1, 1), 64, 128, iters=20)
    run_wgrad("mid1.dense wgrad 64->64", 0, (4, 200, 400), (4, 200, 400), (1, 1, 1), (1, 1, 1), (0, 0, 0), 64, 64, iters=20, in_bn=True)
