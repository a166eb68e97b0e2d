"""The middle layers' contractions alone (forward and data gradient), Lyft grid (GPU box only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import run

if __name__ == "__main__":
    run("mid1 fwd  s(2,1,1) p1", 0, (8, 200, 400), (4, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, iters=20)
    run("mid2 fwd  s1 p(0,1,1)", 0, (4, 200, 400), (2, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64, iters=20)
    run("mid3 fwd  s(2,1,1) p1", 0, (2, 200, 400), (1, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, iters=20)
    run("mid2 dgrad", 1, (2, 200, 400), (4, 200, 400), (3, 3, 3), (1, 1, 1), (0, 1, 1), 64, 64, iters=20)
    run("mid3 dgrad", 1, (1, 200, 400), (2, 200, 400), (3, 3, 3), (2, 1, 1), (1, 1, 1), 64, 64, iters=20)
    run("rpn1 conv 128->128", 0, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, iters=20, in_bn=True)
    run("rpn1 dgrad 128->128", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 128, iters=20)
    run("up1 fwd 128->256 (transposed)", 1, (1, 100, 200), (1, 100, 200), (1, 3, 3), (1, 1, 1), (0, 1, 1), 128, 256, iters=20, in_bn=True)
