"""Cross-check helper run by an interpreter that HAS h5py (this image: /opt/conda/bin/python3.9, h5py 3.3.0 on
libhdf5 1.10.6).  It imports nothing from the repository -- it is the independent side of the comparison in
tests/test_hdf5_lite.py and the generator of tests/golden/keras_layout_*.h5.

    python3.9 h5py_probe.py dump FILE            -> JSON description of every object / attribute in FILE
    python3.9 h5py_probe.py make SPEC.npz FILE [earliest|latest|chunked]
         writes the arrays of SPEC.npz the way Keras' save_model_to_hdf5 / save_weights_to_hdf5_group lay a
         model out: root attrs keras_version / backend / model_config / training_config, group model_weights
         with attrs layer_names / backend / keras_version, one group per layer with attr weight_names and
         the datasets <layer>/<weight>:0.  SPEC.npz keys are "<layer>|<weight name>"; "__config__" holds the
         model_config JSON as uint8.
"""
import hashlib
import json
import sys

import h5py
import numpy as np


def describe_value(v):
    if isinstance(v, bytes):
        return {"kind": "bytes", "value": v.decode("latin1")}
    if isinstance(v, str):
        return {"kind": "str", "value": v}
    a = np.asarray(v)
    if a.dtype.kind == "S":
        return {"kind": "bytes_array", "shape": list(a.shape), "value": [x.decode("latin1") for x in a.ravel()]}
    if a.dtype.kind in "OU":
        return {"kind": "str_array", "shape": list(a.shape), "value": [str(x) for x in a.ravel()]}
    return {"kind": "num", "shape": list(a.shape), "dtype": a.dtype.str,
            "sha": hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()}


def dump(path):
    out = {}
    with h5py.File(path, "r") as f:
        def visit(name, obj):
            d = {"attrs": {k: describe_value(v) for k, v in obj.attrs.items()}}
            if isinstance(obj, h5py.Dataset):
                a = obj[()]
                d.update(type="dataset", shape=list(obj.shape), dtype=obj.dtype.str,
                         sha=hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest())
            else:
                d.update(type="group", keys=sorted(obj.keys()))
            out["/" + name] = d
        out["/"] = {"type": "group", "keys": sorted(f.keys()),
                    "attrs": {k: describe_value(v) for k, v in f.attrs.items()}}
        f.visititems(visit)
    json.dump(out, sys.stdout)


def make(spec_path, path, variant):
    spec = np.load(spec_path, allow_pickle=False)
    layers = {}
    for key in spec.files:
        if key == "__config__":
            continue
        layer, wname = key.split("|")
        layers.setdefault(layer, []).append((wname, spec[key]))
    kwargs = {}
    if variant == "latest":
        # version-2 object headers, compact link messages; more than 8 links per group would switch libhdf5 to
        # dense (fractal heap) link storage, which Keras files never have and hdf5_lite does not read
        kwargs = {"libver": "latest"}
        layers = dict(list(layers.items())[:6])
    with h5py.File(path, "w", **kwargs) as f:
        f.attrs["keras_version"] = "2.4.0"                       # str -> variable-length UTF-8 string
        f.attrs["backend"] = "tensorflow"
        f.attrs["model_config"] = bytes(spec["__config__"])       # bytes -> fixed-length string
        f.attrs["training_config"] = json.dumps({"loss": ["mse", "mse"]})
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = [n.encode("utf8") for n in layers]
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.4.0"
        for layer, weights in layers.items():
            lg = g.create_group(layer)
            lg.attrs["weight_names"] = [w.encode("utf8") for w, _ in weights if w]
            for wname, val in weights:
                if not wname:
                    continue                                   # a layer without weights (e.g. "re_lu|")
                if variant == "chunked" and val.ndim >= 1 and val.size >= 4:
                    chunks = tuple(max(1, (s + 1) // 2) for s in val.shape)
                    d = lg.create_dataset(wname, val.shape, dtype=val.dtype, chunks=chunks, compression="gzip",
                                          shuffle=True, fletcher32=True)
                else:
                    d = lg.create_dataset(wname, val.shape, dtype=val.dtype)
                if val.shape:
                    d[:] = val
                else:
                    d[()] = val


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    elif sys.argv[1] == "make":
        make(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "earliest")
    else:
        raise SystemExit("usage: dump FILE | make SPEC.npz FILE [variant]")
