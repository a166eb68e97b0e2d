"""Generates tests/golden/keras_layout_spec.npz and keras_layout_{earliest,latest,chunked}.h5.

The .h5 files are written by REAL h5py/libhdf5 (tests/golden/h5py_probe.py under /opt/conda/bin/python3.9) in the
layout Keras uses, so that lisec_amd.hdf5_lite's reader is pinned against libhdf5's writer even where that
interpreter is missing.  Run from the repository root:  python tests/golden/make_h5_goldens.py
"""
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
H5PY_PYTHON = "/opt/conda/bin/python3.9"


def spec_arrays(seed=7):
    rng = np.random.default_rng(seed)
    spec = {}
    layers = ["InputVoxel", "reshape", "dense", "reshape_1", "batch_normalization", "activation",
              "max_pooling_vfe_layer", "repeat_layer", "concatenate", "zero_padding3d", "conv3d", "conv2d",
              "conv2d_transpose", "ClassificationLayer"]
    for name in layers:
        if name == "dense":
            spec[f"{name}|{name}/kernel:0"] = rng.standard_normal((6, 16)).astype(np.float32)
        elif name == "batch_normalization":
            for w in ("gamma", "beta", "moving_mean", "moving_variance"):
                spec[f"{name}|{name}/{w}:0"] = rng.standard_normal(16).astype(np.float32)
        elif name == "conv3d":
            spec[f"{name}|{name}/kernel:0"] = rng.standard_normal((3, 3, 3, 4, 5)).astype(np.float32)
            spec[f"{name}|{name}/bias:0"] = rng.standard_normal(5).astype(np.float32)
        elif name in ("conv2d", "ClassificationLayer"):
            spec[f"{name}|{name}/kernel:0"] = rng.standard_normal((3, 3, 5, 7)).astype(np.float32)
            spec[f"{name}|{name}/bias:0"] = rng.standard_normal(7).astype(np.float64)
        elif name == "conv2d_transpose":
            spec[f"{name}|{name}/kernel:0"] = rng.standard_normal((2, 2, 9, 7)).astype(np.float32)
            spec[f"{name}|{name}/iter:0"] = np.array(180, dtype=np.int64)
        else:
            spec[f"{name}|"] = np.zeros(1, np.float32)
    cfg = {"class_name": "Functional", "config": {"name": "model", "layers": [
        {"class_name": n, "config": {"name": n, "note": "x" * 150}} for n in layers]}}
    spec["__config__"] = np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8)
    return spec


if __name__ == "__main__":
    spec_path = os.path.join(HERE, "keras_layout_spec.npz")
    np.savez(spec_path, **spec_arrays())
    for variant in ("earliest", "latest", "chunked"):
        out = os.path.join(HERE, f"keras_layout_{variant}.h5")
        env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
        subprocess.run([H5PY_PYTHON, os.path.join(HERE, "h5py_probe.py"), "make", spec_path, out, variant],
                       check=True, env=env)
        print(out, os.path.getsize(out))
