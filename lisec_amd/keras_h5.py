"""Keras HDF5 checkpoint layout for the Lisec network.

The reference persists its model with Keras: model.save(save_path) (model_training.py:302) and
load_model(path, custom_objects={'RepeatLayer':..., 'MaxPoolingVFELayer':...}) (model_training.py:337-338,
Predict.py:51-52).  A Keras `.h5` holds

    /                   attrs keras_version, backend, model_config (JSON), training_config (JSON)
    /model_weights      attrs layer_names, backend, keras_version
        /<layer>        attr weight_names = [b'<layer>/kernel:0', ...]; datasets at <layer>/<weight>:0
    /optimizer_weights  attr weight_names = [b'SGD/iter:0', b'SGD/<layer>/<weight>/momentum:0', ...] + datasets

This module rebuilds the layer list that createModel (model_training.py:222-257) produces -- with the names Keras
assigns automatically (dense, dense_1, batch_normalization_7, conv2d_transpose_2 ...) and the order of
Model.layers (decreasing depth, ties in traversal order) -- maps every Keras variable onto the flat ParamStore
names of lisec_amd.params, and reads / writes the file through lisec_amd.hdf5_lite.

PARITY UNPINNED against Keras itself (TensorFlow is not installed here): the HDF5 container is cross-checked
against h5py/libhdf5 in tests/test_hdf5_lite.py; the Keras naming rules are restated from Keras' public behaviour.
"""
import json
import re

import numpy as np

from . import hdf5_lite
from .params import DECONVS, MID, RPN_BLOCKS

KERAS_VERSION = "2.4.0"
BN_WEIGHTS = ("gamma", "beta", "moving_mean", "moving_variance")


def _snake(cls):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", cls)
    return re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()


class _Graph:
    """Records layers as createModel creates them; names follow Keras' per-class counters."""

    def __init__(self):
        self.layers, self.counts = [], {}

    def add(self, cls, inbound, config=None, weights=(), name=None):
        if name is None:
            base = _snake(cls)
            n = self.counts.get(base, 0)
            self.counts[base] = n + 1
            name = base if n == 0 else f"{base}_{n}"
        cfg = {"name": name, "trainable": True, "dtype": "float32"}
        cfg.update(config or {})
        self.layers.append(dict(name=name, class_name=cls, config=cfg,
                                inbound=[inbound] if isinstance(inbound, str) else list(inbound),
                                weights=list(weights)))
        return name


def _glorot():
    return {"class_name": "GlorotUniform", "config": {"seed": None}}


def _zeros():
    return {"class_name": "Zeros", "config": {}}


def keras_layers(nx, ny, nz, maxPoints):
    """[dict(name, class_name, config, inbound, weights=[(keras weight, ParamStore name)])] in Model.layers order,
    plus the output layer names."""
    g = _Graph()
    T = maxPoints
    shape = (nz, nx, ny, T, 6)
    x = g.add("InputLayer", [], {"batch_input_shape": [None] + list(shape), "sparse": False, "ragged": False},
              name="InputVoxel")
    del g.layers[-1]["config"]["trainable"]

    def dense(x, shape, units, act, pname):                              # addDenseLayer :177-185
        flat = (int(np.prod(shape[:-2])),) + tuple(shape[-2:])
        x = g.add("Reshape", x, {"target_shape": list(flat)})
        x = g.add("Dense", x, {"units": units, "activation": act or "linear", "use_bias": False,
                               "kernel_initializer": _glorot(), "bias_initializer": _zeros(),
                               "batch_input_shape": [None] + list(flat)},
                  weights=[("kernel", pname + ".kernel")])
        shape = tuple(shape[:-1]) + (units,)
        x = g.add("Reshape", x, {"target_shape": list(shape)})
        return x, shape

    def bn(x, rank, pname):
        return g.add("BatchNormalization", x, {"axis": [rank], "momentum": 0.99, "epsilon": 0.001, "center": True,
                                               "scale": True},
                     weights=[(w, f"{pname}.{w}") for w in BN_WEIGHTS])

    def fcn(x, shape, units, pname):                                     # addFCN :168-173
        x, shape = dense(x, shape, units, None, pname + ".dense")
        x = bn(x, len(shape), pname + ".bn")
        x = g.add("Activation", x, {"activation": "relu"})
        return x, shape

    def vfe(x, shape, units, pname):                                     # addVFELayer :155-165
        x, shape = fcn(x, shape, units // 2, pname)
        pool = g.add("MaxPoolingVFELayer", x, {"combine": False})
        rep = g.add("RepeatLayer", pool)
        x = g.add("Concatenate", [rep, x], {"axis": -1})
        return x, tuple(shape[:-1]) + (units,)

    x, shape = vfe(x, shape, 32, "vfe1")
    x, shape = vfe(x, shape, 64, "vfe2")
    x, shape = fcn(x, shape, 64, "fcn")
    x = g.add("MaxPoolingVFELayer", x, {"combine": True})
    shape = tuple(shape[:3]) + (64,)
    for i, (stride, pad) in enumerate(MID):                              # addConv3DLayer :190-195
        x = g.add("ZeroPadding3D", x, {"padding": [[p, p] for p in pad], "data_format": "channels_last"})
        x = g.add("Conv3D", x, {"filters": 64, "kernel_size": [3, 3, 3], "strides": list(stride), "padding": "valid",
                                "data_format": "channels_last", "dilation_rate": [1, 1, 1], "groups": 1,
                                "activation": "linear", "use_bias": True, "kernel_initializer": _glorot(),
                                "bias_initializer": _zeros()},
                  weights=[("kernel", f"mid{i+1}.conv.kernel"), ("bias", f"mid{i+1}.conv.bias")])
        shape = tuple((s + 2 * p - 3) // st + 1 for s, p, st in zip(shape[:3], pad, stride)) + (64,)
        x = bn(x, 4, f"mid{i+1}.bn")
        x, shape = dense(x, shape, 64, "relu", f"mid{i+1}.dense")
    x = g.add("Permute", x, {"dims": [2, 3, 4, 1]})
    shape = (shape[1], shape[2], shape[3] * shape[0])
    x = g.add("Reshape", x, {"target_shape": list(shape)})

    def conv2d(x, cout, stride, pname_conv, pname_bn):                   # addConv2DLayer :200-206
        x = g.add("ZeroPadding2D", x, {"padding": [[1, 1], [1, 1]], "data_format": "channels_last"})
        x = g.add("Conv2D", x, {"filters": cout, "kernel_size": [3, 3], "strides": [stride, stride],
                                "padding": "valid", "data_format": "channels_last", "dilation_rate": [1, 1],
                                "groups": 1, "activation": "linear", "use_bias": True,
                                "kernel_initializer": _glorot(), "bias_initializer": _zeros()},
                  weights=[("kernel", pname_conv + ".kernel"), ("bias", pname_conv + ".bias")])
        x = bn(x, 3, pname_bn)
        return g.add("Activation", x, {"activation": "relu"})

    ups = []
    for b, (cout, q) in enumerate(RPN_BLOCKS):                           # addRPNConvLayer :210-214, :245-252
        for j in range(q + 1):
            x = conv2d(x, cout, 2 if j == 0 else 1, f"rpn{b+1}.conv{j}", f"rpn{b+1}.bn{j}")
        k, s = DECONVS[b]
        ups.append(g.add("Conv2DTranspose", x, {
            "filters": 256, "kernel_size": [k, k], "strides": [s, s], "padding": "same",
            "data_format": "channels_last", "dilation_rate": [1, 1], "groups": 1, "activation": "linear",
            "use_bias": True, "output_padding": None, "kernel_initializer": _glorot(), "bias_initializer": _zeros()},
            weights=[("kernel", f"up{b+1}.kernel"), ("bias", f"up{b+1}.bias")]))
    cat = g.add("Concatenate", ups, {"axis": -1})
    outs = []
    for name, filters, p in (("ClassificationLayer", 2, "cls"), ("RegressionLayer", 14, "reg")):
        outs.append(g.add("Conv2D", cat, {"filters": filters, "kernel_size": [1, 1], "strides": [1, 1],
                                          "padding": "same", "data_format": "channels_last", "dilation_rate": [1, 1],
                                          "groups": 1, "activation": "linear", "use_bias": True,
                                          "kernel_initializer": _glorot(), "bias_initializer": _zeros()},
                          weights=[("kernel", p + ".kernel"), ("bias", p + ".bias")], name=name))
    return _model_layers_order(g.layers, outs), outs


def _model_layers_order(layers, outputs):
    """Order of keras Model.layers: decreasing depth (longest path to an output); equal depths in the order a
    depth-first walk from the outputs first completes them (functional.py _map_graph_network)."""
    by_name = {L["name"]: L for L in layers}
    depth, index = {}, {}

    def visit(name):                                   # post-order: inputs before the layer itself
        stack = [(name, iter(by_name[name]["inbound"]))]
        seen = {name}
        while stack:
            cur, it = stack[-1]
            nxt = next(it, None)
            if nxt is None:
                stack.pop()
                if cur not in index:
                    index[cur] = len(index)
            elif nxt not in index and nxt not in seen:
                seen.add(nxt)
                stack.append((nxt, iter(by_name[nxt]["inbound"])))
    for o in outputs:
        visit(o)
    # longest path to an output: relax consumers before producers (reverse creation order is a valid order)
    for L in layers:
        depth[L["name"]] = 0
    for L in reversed(layers):
        for src in L["inbound"]:
            depth[src] = max(depth[src], depth[L["name"]] + 1)
    return sorted(layers, key=lambda L: (-depth[L["name"]], index[L["name"]]))


def model_config(nx, ny, nz, maxPoints):
    layers, outs = keras_layers(nx, ny, nz, maxPoints)
    cfg_layers = []
    for L in layers:
        inbound = [[[src, 0, 0, {}] for src in L["inbound"]]] if L["inbound"] else []
        cfg_layers.append({"class_name": L["class_name"], "config": L["config"], "name": L["name"],
                           "inbound_nodes": inbound})
    return {"class_name": "Functional",
            "config": {"name": "model", "layers": cfg_layers, "input_layers": [["InputVoxel", 0, 0]],
                       "output_layers": [[o, 0, 0] for o in outs]},
            "keras_version": KERAS_VERSION, "backend": "tensorflow"}


def _training_config(optimizer):
    o = optimizer or {}
    return {"loss": ["mse", "mse"], "metrics": None, "weighted_metrics": None, "loss_weights": None,
            "optimizer_config": {"class_name": "SGD", "config": {
                "name": "SGD", "learning_rate": float(o.get("lr", 0.01)), "decay": float(o.get("decay", 0.0)),
                "momentum": float(o.get("momentum", 0.0)), "nesterov": bool(o.get("nesterov", False))}}}


# ---------------------------------------------------------------------------------------------------
def save_model(path, params, nx, ny, nz, maxPoints, optimizer=None, iterations=0, velocity=None):
    """params: dict ParamStore name -> array.  optimizer: dict(lr, decay, momentum, nesterov) or None (a model that
    was never compiled: no training_config / optimizer_weights, like Keras).  velocity: dict of trainable
    ParamStore name -> momentum accumulator."""
    layers, _ = keras_layers(nx, ny, nz, maxPoints)
    with hdf5_lite.File(path, "w") as f:
        f.attrs["keras_version"] = KERAS_VERSION.encode()
        f.attrs["backend"] = b"tensorflow"
        f.attrs["model_config"] = json.dumps(model_config(nx, ny, nz, maxPoints)).encode("utf8")
        if optimizer is not None:
            f.attrs["training_config"] = json.dumps(_training_config(optimizer)).encode("utf8")
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = [L["name"].encode("utf8") for L in layers]
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = KERAS_VERSION.encode()
        slots = []
        for L in layers:
            lg = g.create_group(L["name"])
            names = [f"{L['name']}/{w}:0" for w, _ in L["weights"]]
            lg.attrs["weight_names"] = [n.encode("utf8") for n in names]
            for n, (w, pname) in zip(names, L["weights"]):
                lg.create_dataset(n, data=np.asarray(params[pname], dtype=np.float32))
                if w in ("kernel", "bias", "gamma", "beta"):
                    slots.append((f"SGD/{L['name']}/{w}/momentum:0", pname))
        if optimizer is not None and velocity is not None:
            og = f.create_group("optimizer_weights")
            og.attrs["weight_names"] = [b"SGD/iter:0"] + [n.encode("utf8") for n, _ in slots]
            og.create_dataset("SGD/iter:0", data=np.array(int(iterations), dtype=np.int64))
            for n, pname in slots:
                og.create_dataset(n, data=np.asarray(velocity[pname], dtype=np.float32))


def _text(v):
    if isinstance(v, (bytes, np.bytes_)):
        return bytes(v).decode("utf8")
    return str(v)


def _attr_list(group, name):
    """load_attributes_from_hdf5_group: an attribute too large for one object header message is split into
    name0, name1, ... by Keras."""
    if name in group.attrs:
        return [_text(x) for x in np.asarray(group.attrs[name]).ravel()]
    out, i = [], 0
    while f"{name}{i}" in group.attrs:
        out.extend(_text(x) for x in np.asarray(group.attrs[f"{name}{i}"]).ravel())
        i += 1
    return out


_BASES = ("batch_normalization", "conv2d_transpose", "conv3d", "conv2d", "dense", "ClassificationLayer",
          "RegressionLayer")


def _base_of(name):
    return next((b for b in _BASES if name == b or re.fullmatch(re.escape(b) + r"_\d+", name)), None)


def _suffix_number(name, base):
    rest = name[len(base):]
    return int(rest[1:]) if rest else 0


def load_model(path, grid=None):
    """Reads a Keras `.h5` written by the reference (or by save_model).  Returns dict(params, nx, ny, nz, maxPoints,
    iterations, velocity (dict or None), optimizer (dict or None)).

    Layers are matched by class and creation order (the numeric suffix of Keras' automatic names), not by the exact
    suffix: a model built as the second one of a Python session carries shifted suffixes."""
    with hdf5_lite.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        if "layer_names" not in g.attrs and "layer_names0" not in g.attrs:
            raise hdf5_lite.H5Error(f"{path}: no Keras layer_names attribute")
        cfg = json.loads(_text(f.attrs["model_config"])) if "model_config" in f.attrs else None
        if cfg is not None:
            inp = next(L for L in cfg["config"]["layers"] if L["class_name"] == "InputLayer")
            _, nz, nx, ny, T, _ = inp["config"]["batch_input_shape"]
        elif grid is not None:
            nx, ny, nz, T = grid
        else:
            from . import Constants
            nx, ny, nz, T = Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints
        found = {}                                    # class base -> [(order, layer name, {weight kind: array})]
        for lname in _attr_list(g, "layer_names"):
            wnames = _attr_list(g[lname], "weight_names")
            if not wnames:
                continue
            base = _base_of(lname)
            if base is None:
                raise hdf5_lite.H5Error(f"{path}: layer {lname!r} with weights is not part of the Lisec network")
            vals = {w.split("/")[-1].split(":")[0]: np.asarray(g[lname][w][()], dtype=np.float32) for w in wnames}
            found.setdefault(base, []).append((_suffix_number(lname, base), lname, vals))
        for v in found.values():
            v.sort(key=lambda t: t[0])
        layers, _ = keras_layers(nx, ny, nz, T)
        want = {}
        for L in layers:
            if L["weights"]:
                base = _base_of(L["name"])
                want.setdefault(base, []).append((_suffix_number(L["name"], base), L))
        params, keras_to_param = {}, {}
        for base, lst in want.items():
            lst.sort(key=lambda t: t[0])
            have = found.get(base, [])
            if len(have) != len(lst):
                raise hdf5_lite.H5Error(f"{path}: {len(have)} {base} layers with weights, the network has {len(lst)}")
            for (_, L), (_, lname, vals) in zip(lst, have):
                for w, pname in L["weights"]:
                    if w not in vals:
                        raise hdf5_lite.H5Error(f"{path}: layer {lname} lacks {w}")
                    params[pname] = vals[w]
                    keras_to_param[f"{lname}/{w}"] = pname
        out = dict(params=params, nx=int(nx), ny=int(ny), nz=int(nz), maxPoints=int(T), iterations=0, velocity=None,
                   optimizer=None)
        if "training_config" in f.attrs:
            oc = json.loads(_text(f.attrs["training_config"])).get("optimizer_config", {})
            c = oc.get("config", {})
            if oc.get("class_name") == "SGD":
                out["optimizer"] = dict(lr=c.get("learning_rate", c.get("lr", 0.01)), decay=c.get("decay", 0.0),
                                        momentum=c.get("momentum", 0.0), nesterov=c.get("nesterov", False))
        if "optimizer_weights" in f:
            og = f["optimizer_weights"]
            vel = {}
            for w in _attr_list(og, "weight_names"):
                a = og[w][()]
                if w.endswith("iter:0"):
                    out["iterations"] = int(a)
                    continue
                m = re.fullmatch(r"[^/]+/(.+)/momentum:0", w)
                if m and m.group(1) in keras_to_param:
                    vel[keras_to_param[m.group(1)]] = np.asarray(a, dtype=np.float32)
            if vel:
                out["velocity"] = vel
        return out
