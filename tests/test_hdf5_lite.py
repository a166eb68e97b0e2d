"""lisec_amd.hdf5_lite / keras_h5 (Keras `.h5` checkpoints without libhdf5).

Pinned three ways: (1) files written by REAL h5py/libhdf5 in Keras' layout (tests/golden/keras_layout_*.h5, made by
tests/golden/make_h5_goldens.py) are read back and compared with the arrays they were made from; (2) where an
interpreter with h5py exists (/opt/conda/bin/python3.9 in this image) files written by hdf5_lite are read by h5py
and h5py-written files by hdf5_lite, live; (3) write -> read round trips of the full Lisec checkpoint.
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from lisec_amd import hdf5_lite, keras_h5
from lisec_amd.params import glorot_numpy, param_specs, TRAINABLE_KINDS

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
H5PY_PYTHON = "/opt/conda/bin/python3.9"
PROBE = os.path.join(GOLDEN, "h5py_probe.py")


def _have_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    return subprocess.run([H5PY_PYTHON, "-c", "import h5py"], env=env, capture_output=True).returncode == 0


def _probe(*args):
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    return subprocess.run([H5PY_PYTHON, PROBE, *args], env=env, capture_output=True, check=True).stdout


def _text(v):
    return bytes(v).decode() if isinstance(v, (bytes, np.bytes_)) else str(v)


@pytest.mark.parametrize("variant", ["earliest", "latest", "chunked"])
def test_reads_h5py_written_keras_layout(variant):
    """earliest: superblock 0, v1 headers, symbol-table groups (what Keras files are); latest: superblock 3, v2 headers,
    compact links; chunked: gzip + shuffle + fletcher32 chunks behind a v1 B-tree."""
    spec = np.load(os.path.join(GOLDEN, "keras_layout_spec.npz"))
    with hdf5_lite.File(os.path.join(GOLDEN, f"keras_layout_{variant}.h5")) as f:
        assert _text(f.attrs["keras_version"]) == "2.4.0" and _text(f.attrs["backend"]) == "tensorflow"
        assert _text(f.attrs["model_config"]) == bytes(spec["__config__"]).decode()
        assert json.loads(_text(f.attrs["training_config"])) == {"loss": ["mse", "mse"]}
        g = f["model_weights"]
        layers = [_text(n) for n in g.attrs["layer_names"]]
        assert sorted(g.keys()) == sorted(layers)
        assert len(layers) == (6 if variant == "latest" else 14)
        checked = 0
        for key in spec.files:
            if key == "__config__":
                continue
            layer, w = key.split("|")
            if layer not in layers:
                continue
            names = [_text(n) for n in np.asarray(g[layer].attrs["weight_names"]).ravel()]
            if not w:
                assert names == [] and g[layer].keys() == []
                continue
            assert w in names
            d = g[layer][w]
            a = d[()]
            assert d.shape == spec[key].shape and a.dtype == spec[key].dtype
            np.testing.assert_array_equal(a, spec[key])
            np.testing.assert_array_equal(f[f"/model_weights/{layer}/{w}"][()], spec[key])     # absolute path
            checked += 1
        assert checked >= 2
        with pytest.raises(KeyError):
            g["no_such_layer"]


def test_rejects_non_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"PK\x03\x04" + bytes(600))
    with pytest.raises(hdf5_lite.H5Error):
        hdf5_lite.File(str(p))


def _write_sample(path, n_layers=300):
    rng = np.random.default_rng(3)
    expect = {}
    with hdf5_lite.File(path, "w") as f:
        f.attrs["keras_version"] = b"2.4.0"
        f.attrs["backend"] = "tensorflow"
        f.attrs["model_config"] = json.dumps({"pad": "x" * 40000}).encode()
        g = f.create_group("model_weights")
        names = [f"layer_{i}".encode() for i in range(n_layers)]
        g.attrs["layer_names"] = names
        g.attrs["nums"] = np.arange(5, dtype=np.int64)
        g.attrs["pi"] = np.float32(3.25)
        for i, n in enumerate(names):
            n = n.decode()
            lg = g.create_group(n)
            if i % 2:
                lg.attrs["weight_names"] = []
                continue
            lg.attrs["weight_names"] = [f"{n}/kernel:0".encode(), f"{n}/bias:0".encode()]
            k = rng.standard_normal((3, 3, 4, 5)).astype(np.float32)
            b = rng.standard_normal(5)
            lg.create_dataset(f"{n}/kernel:0", data=k)
            lg.create_dataset(f"{n}/bias:0", data=b)
            expect[f"/model_weights/{n}/{n}/kernel:0"] = k
            expect[f"/model_weights/{n}/{n}/bias:0"] = b
        f.create_dataset("ints", data=np.arange(12, dtype=np.int32).reshape(3, 4))
        f.create_dataset("scalar", data=np.int64(180))
        f.create_group("empty")
    expect["/ints"] = np.arange(12, dtype=np.int32).reshape(3, 4)
    expect["/scalar"] = np.array(180, dtype=np.int64)
    return expect


def test_write_read_roundtrip(tmp_path):
    """300 layer groups: 38 symbol-table nodes under a two-level group B-tree."""
    path = str(tmp_path / "rt.h5")
    expect = _write_sample(path)
    with hdf5_lite.File(path) as f:
        assert sorted(f.keys()) == ["empty", "ints", "model_weights", "scalar"]
        assert f["empty"].keys() == []
        assert f.attrs["keras_version"] == b"2.4.0" and f.attrs["backend"] == b"tensorflow"
        assert len(f.attrs["model_config"]) > 40000
        g = f["model_weights"]
        assert len(g.keys()) == 300 and [n.decode() for n in g.attrs["layer_names"]][7] == "layer_7"
        np.testing.assert_array_equal(g.attrs["nums"], np.arange(5))
        assert g.attrs["pi"] == np.float32(3.25) and np.asarray(g.attrs["pi"]).shape == ()
        assert len(g["layer_1"].attrs["weight_names"]) == 0
        for k, v in expect.items():
            a = f[k][()]
            assert a.dtype == v.dtype
            np.testing.assert_array_equal(a, v)
        assert int(f["scalar"][()]) == 180


def test_oversized_attribute_is_refused(tmp_path):
    with pytest.raises(hdf5_lite.H5Error):
        with hdf5_lite.File(str(tmp_path / "big.h5"), "w") as f:
            f.attrs["model_config"] = b"x" * 70000


@pytest.mark.skipif(not _have_h5py(), reason="no interpreter with h5py in this image")
def test_h5py_reads_what_hdf5_lite_writes(tmp_path):
    path = str(tmp_path / "mine.h5")
    expect = _write_sample(path, n_layers=70)
    desc = json.loads(_probe("dump", path))
    assert desc["/"]["keys"] == ["empty", "ints", "model_weights", "scalar"]
    assert desc["/"]["attrs"]["keras_version"] == {"kind": "bytes", "value": "2.4.0"}
    assert len(desc["/"]["attrs"]["model_config"]["value"]) > 40000
    mw = desc["/model_weights"]
    assert len(mw["keys"]) == 70 and mw["attrs"]["layer_names"]["value"][3] == "layer_3"
    assert mw["attrs"]["pi"]["shape"] == [] and mw["attrs"]["nums"]["shape"] == [5]
    assert desc["/model_weights/layer_0"]["attrs"]["weight_names"]["value"] == ["layer_0/kernel:0", "layer_0/bias:0"]
    for k, v in expect.items():
        d = desc[k]
        assert d["type"] == "dataset" and tuple(d["shape"]) == v.shape and np.dtype(d["dtype"]) == v.dtype
        assert d["sha"] == hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest(), k


@pytest.mark.skipif(not _have_h5py(), reason="no interpreter with h5py in this image")
def test_hdf5_lite_reads_what_h5py_writes_live(tmp_path):
    """Full-size weights: the real Lisec variables written by h5py in Keras' layout, read through keras_h5."""
    params = glorot_numpy(seed=11)
    rng = np.random.default_rng(5)
    for name, shape, kind in param_specs():
        if kind != "kernel":
            params[name] = rng.standard_normal(shape).astype(np.float32)
    layers, _ = keras_h5.keras_layers(200, 400, 8, 35)
    spec = {}
    for L in layers:
        if not L["weights"]:
            spec[f"{L['name']}|"] = np.zeros(1, np.float32)
        for w, pname in L["weights"]:
            spec[f"{L['name']}|{L['name']}/{w}:0"] = params[pname]
    spec["__config__"] = np.frombuffer(json.dumps(keras_h5.model_config(200, 400, 8, 35)).encode(), dtype=np.uint8)
    spec_path, h5_path = str(tmp_path / "spec.npz"), str(tmp_path / "keras.h5")
    np.savez(spec_path, **spec)
    _probe("make", spec_path, h5_path, "earliest")
    ck = keras_h5.load_model(h5_path)
    assert (ck["nx"], ck["ny"], ck["nz"], ck["maxPoints"]) == (200, 400, 8, 35)
    assert set(ck["params"]) == set(params)
    for k in params:
        np.testing.assert_array_equal(ck["params"][k], params[k])


def test_keras_names_and_order():
    layers, outs = keras_h5.keras_layers(200, 400, 8, 35)
    names = [L["name"] for L in layers]
    assert len(names) == len(set(names)) == 113 and outs == ["ClassificationLayer", "RegressionLayer"]
    assert names[:9] == ["InputVoxel", "reshape", "dense", "reshape_1", "batch_normalization", "activation",
                         "max_pooling_vfe_layer", "repeat_layer", "concatenate"]
    # Model.layers order: the three Conv2DTranspose layers share a depth and follow every RPN block
    assert names[-6:] == ["conv2d_transpose", "conv2d_transpose_1", "conv2d_transpose_2", "concatenate_2",
                          "ClassificationLayer", "RegressionLayer"]
    assert names.index("conv2d_15") < names.index("conv2d_transpose")
    count = lambda base: sum(1 for n in names if keras_h5._base_of(n) == base)
    assert (count("dense"), count("batch_normalization"), count("conv3d"), count("conv2d"),
            count("conv2d_transpose")) == (6, 22, 3, 16, 3)
    by = {L["name"]: L for L in layers}
    assert by["dense_3"]["weights"] == [("kernel", "mid1.dense.kernel")]
    assert by["batch_normalization_6"]["weights"][0] == ("gamma", "rpn1.bn0.gamma")
    assert by["conv2d_4"]["weights"][0] == ("kernel", "rpn2.conv0.kernel")
    assert by["reshape"]["config"]["target_shape"] == [8 * 200 * 400, 35, 6]
    assert by["reshape_12"]["config"]["target_shape"] == [200, 400, 64]
    mapped = [p for L in layers for _, p in L["weights"]]
    assert sorted(mapped) == sorted(n for n, _, _ in param_specs())
    cfg = keras_h5.model_config(200, 400, 8, 35)
    assert len(json.dumps(cfg)) < 60000                    # must fit one object-header message
    assert cfg["config"]["layers"][8]["inbound_nodes"] == [[["repeat_layer", 0, 0, {}], ["activation", 0, 0, {}]]]


def test_lisec_checkpoint_roundtrip(tmp_path):
    params = glorot_numpy(seed=2)
    rng = np.random.default_rng(9)
    for name, shape, kind in param_specs():
        if kind != "kernel":
            params[name] = rng.standard_normal(shape).astype(np.float32)
    vel = {n: rng.standard_normal(s).astype(np.float32) for n, s, k in param_specs() if k in TRAINABLE_KINDS}
    path = str(tmp_path / "lisec.h5")
    keras_h5.save_model(path, params, 200, 400, 8, 35, optimizer=dict(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True),
                        iterations=180, velocity=vel)
    ck = keras_h5.load_model(path)
    assert (ck["nx"], ck["ny"], ck["nz"], ck["maxPoints"], ck["iterations"]) == (200, 400, 8, 35, 180)
    assert ck["optimizer"] == dict(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True)
    for k in params:
        np.testing.assert_array_equal(ck["params"][k], params[k])
    assert set(ck["velocity"]) == set(vel)
    for k in vel:
        np.testing.assert_array_equal(ck["velocity"][k], vel[k])
    # a model built second in a Python session carries shifted automatic names: matched by order, not by suffix
    with hdf5_lite.File(path) as f:
        names = [n.decode() for n in f["model_weights"].attrs["layer_names"]]
    shifted = str(tmp_path / "shifted.h5")
    with hdf5_lite.File(path) as src, hdf5_lite.File(shifted, "w") as dst:
        dst.attrs["model_config"] = src.attrs["model_config"]
        g = dst.create_group("model_weights")
        ren = {}
        for n in names:
            base = keras_h5._base_of(n)
            if base in ("dense", "batch_normalization", "conv2d"):
                ren[n] = f"{base}_{keras_h5._suffix_number(n, base) + 40}"
            else:
                ren[n] = n
        g.attrs["layer_names"] = [ren[n].encode() for n in names]
        for n in names:
            lg = g.create_group(ren[n])
            wn = [w.decode() for w in np.asarray(src["model_weights"][n].attrs["weight_names"]).ravel()]
            lg.attrs["weight_names"] = [w.replace(n + "/", ren[n] + "/", 1).encode() for w in wn]
            for w in wn:
                lg.create_dataset(w.replace(n + "/", ren[n] + "/", 1), data=src["model_weights"][n][w][()])
    ck2 = keras_h5.load_model(shifted)
    for k in params:
        np.testing.assert_array_equal(ck2["params"][k], params[k])
    assert ck2["optimizer"] is None and ck2["velocity"] is None
