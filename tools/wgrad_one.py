"""One weight-gradient geometry launched a few times (profiling target).  usage: python tools/wgrad_one.py <case substring> [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_wgrad_plans import CASES, run
for c in CASES:
    if sys.argv[1] in c[0]:
        print(c[0], run(*c, iters=int(sys.argv[2]) if len(sys.argv) > 2 else 4))
