#!/usr/bin/env python3
"""Golden vectors for the pure-numpy label-side helpers, made by RUNNING the reference function.

Run in the build container only:   python tests/golden/make_box_goldens.py

Imports /root/reference/serialize_data.py unmodified, exactly as make_voxel_goldens.py does (the same inert
placeholder modules stand in for the four absent third-party imports; none of them is touched by what runs here),
and calls its fixBoxScaling (serialize_data.py:184-191) on a few shapes, alone and as applied at :217.  The fixture
holds inputs and outputs only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_voxel_goldens import REF, _install_placeholders  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    _install_placeholders()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import serialize_data as ref
    cases = [((1, 7), 100, 200, 200, 400), ((5, 7), 100, 200, 200, 400), ((3, 7), 50, 25, 200, 400),
             ((12, 7), 100, 200, 100, 200)]
    out = {}
    for i, (shape, nx_, ny_, ox, oy) in enumerate(cases):
        out[f"args{i}"] = np.array(list(shape) + [nx_, ny_, ox, oy], dtype=np.int64)
        out[f"mult{i}"] = np.asarray(ref.fixBoxScaling(shape, nx_, ny_, ox, oy), dtype=np.float64)
    rng = np.random.default_rng(0)
    data = rng.normal(0, 10, (5, 7))
    out["data"] = data
    out["fixed"] = data * ref.fixBoxScaling(data.shape, 100, 200, 200, 400)       # the call of serialize_data.py:217
    np.savez(os.path.join(OUT, "box_fixscaling.npz"), **out)
    print("wrote box_fixscaling.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
