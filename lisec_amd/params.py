"""Variables of the Lisec network as two flat device buffers.

Layer list and shapes follow createModel (reference model_training.py:222-257); layouts are the
Keras ones (Dense (in,out); Conv3D (kd,kh,kw,in,out); Conv2D (kh,kw,in,out); Conv2DTranspose
(kh,kw,out,in)).  All trainable variables live in ONE contiguous fp32 buffer `theta`
(6 491 024 floats) so that the gradient of a step is one contiguous buffer too: the data-parallel
all-reduce is a single RCCL call and the SGD update a single kernel.  BatchNormalization moving
statistics live in a second flat buffer `state` (not trained, not all-reduced).
"""
import math

import numpy as np
import torch

RPN_BLOCKS = ((128, 3), (128, 5), (256, 5))      # (filters, q)        model_training.py:245,248,251
DECONVS = ((3, 1), (2, 2), (4, 4))                # (kernel, stride)    model_training.py:246,249,252
MID = (((2, 1, 1), (1, 1, 1)), ((1, 1, 1), (0, 1, 1)), ((2, 1, 1), (1, 1, 1)))   # (stride, pad) :236-238

TRAINABLE_KINDS = ("kernel", "bias", "gamma", "beta")


def fold_depth(nz):
    """Depth left after the three Conv3D layers (model_training.py:236-238): nz = 8 -> 4 -> 2 -> 1.  The RPN input has
    64 * fold_depth(nz) channels after Permute((2,3,4,1)) + Reshape (:242-243)."""
    d = int(nz)
    for stride, pad in MID:
        d = (d + 2 * pad[0] - 3) // stride[0] + 1
        if d < 1:
            raise ValueError(f"nz={nz}: the middle layers leave no depth")
    return d


def param_specs(dprime=1):
    """[(name, shape, kind)] in forward order.  dprime = fold_depth(nz): the first RPN conv reads 64*dprime channels."""
    specs = []

    def bn(prefix, c):
        specs.extend([(prefix + ".gamma", (c,), "gamma"), (prefix + ".beta", (c,), "beta"),
                      (prefix + ".moving_mean", (c,), "moving_mean"),
                      (prefix + ".moving_variance", (c,), "moving_variance")])

    for name, cin, cout in (("vfe1", 6, 16), ("vfe2", 32, 32), ("fcn", 64, 64)):    # :231-233
        specs.append((f"{name}.dense.kernel", (cin, cout), "kernel"))
        bn(f"{name}.bn", cout)
    for i in range(3):                                                                # :236-238
        specs.append((f"mid{i+1}.conv.kernel", (3, 3, 3, 64, 64), "kernel"))
        specs.append((f"mid{i+1}.conv.bias", (64,), "bias"))
        bn(f"mid{i+1}.bn", 64)
        specs.append((f"mid{i+1}.dense.kernel", (64, 64), "kernel"))
    cin = 64 * dprime
    for b, (cout, q) in enumerate(RPN_BLOCKS):                                        # :245-252
        for j in range(q + 1):
            specs.append((f"rpn{b+1}.conv{j}.kernel", (3, 3, cin, cout), "kernel"))
            specs.append((f"rpn{b+1}.conv{j}.bias", (cout,), "bias"))
            bn(f"rpn{b+1}.bn{j}", cout)
            cin = cout
        k, _ = DECONVS[b]
        specs.append((f"up{b+1}.kernel", (k, k, 256, cout), "kernel"))
        specs.append((f"up{b+1}.bias", (256,), "bias"))
    specs.append(("cls.kernel", (1, 1, 768, 2), "kernel"))                            # :254
    specs.append(("cls.bias", (2,), "bias"))
    specs.append(("reg.kernel", (1, 1, 768, 14), "kernel"))                           # :255
    specs.append(("reg.bias", (14,), "bias"))
    return specs


def glorot_numpy(seed=1234, dprime=1):
    """Keras default initialisers: glorot_uniform kernels, zero biases, BN gamma=1, beta=0,
    moving_mean=0, moving_variance=1.  Returns dict name -> float32 numpy array."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, kind in param_specs(dprime):
        if kind == "kernel":
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            limit = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
            a = rng.uniform(-limit, limit, shape)
        elif kind in ("gamma", "moving_variance"):
            a = np.ones(shape)
        else:
            a = np.zeros(shape)
        out[name] = a.astype(np.float32)
    return out


class ParamStore:
    """theta (trainable) + state (BN moving statistics) as flat fp32 device tensors with named views."""

    def __init__(self, device, init=None, dprime=None):
        self.device = device
        if dprime is None:                 # inferred from the first RPN kernel of `init` (64 * dprime input channels)
            dprime = 1
            if init is not None and "rpn1.conv0.kernel" in init:
                dprime = max(1, int(np.asarray(init["rpn1.conv0.kernel"]).shape[2]) // 64)
        self.dprime = int(dprime)
        self.specs = param_specs(self.dprime)
        self.offsets = {}
        nt = ns = 0
        for name, shape, kind in self.specs:
            n = int(np.prod(shape))
            if kind in TRAINABLE_KINDS:
                self.offsets[name] = ("theta", nt, shape)
                nt += (n + 3) // 4 * 4            # keep every variable 16-byte aligned
            else:
                self.offsets[name] = ("state", ns, shape)
                ns += (n + 3) // 4 * 4
        self.n_theta, self.n_state = nt, ns
        self._views = {}
        self.version = 0          # bumped by every external write (load_dict / touch): consumers repack on change
        self.theta = torch.zeros(nt, dtype=torch.float32, device=device)
        self.state = torch.zeros(ns, dtype=torch.float32, device=device)
        self.load_dict(init if init is not None else glorot_numpy(dprime=self.dprime))

    def view(self, name, buf=None):
        """Named view into theta / state (or into `buf`, a buffer laid out like theta).  Views are cached per buffer:
        the schedule asks for the same few hundred views every step."""
        which, off, shape = self.offsets[name]
        base = buf if buf is not None else (self.theta if which == "theta" else self.state)
        key = (name, base.data_ptr())
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = base[off:off + int(np.prod(shape))].view(*shape)
        return v

    def grad_view(self, grad, name):
        assert self.offsets[name][0] == "theta"
        return self.view(name, buf=grad)

    def ptr(self, name, buf=None):
        import ctypes
        which, off, _ = self.offsets[name]
        base = buf if buf is not None else (self.theta if which == "theta" else self.state)
        return ctypes.c_void_p(base.data_ptr() + 4 * off)

    def load_dict(self, d):
        for name, shape, _ in self.specs:
            a = d[name]
            if isinstance(a, torch.Tensor):
                a = a.detach().cpu().numpy()
            a = np.asarray(a, dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f"{name}: shape {a.shape} != {shape}")
            self.view(name).copy_(torch.from_numpy(np.ascontiguousarray(a)))
        self.version += 1

    def touch(self):
        """Call after writing theta / state directly (e.g. a broadcast into them): invalidates every packed copy."""
        self.version += 1

    def to_dict(self):
        return {name: self.view(name).detach().cpu().numpy().copy() for name, _, _ in self.specs}

    def trainable_names(self):
        return [n for n, _, k in self.specs if k in TRAINABLE_KINDS]

    def n_trainable(self):
        return sum(int(np.prod(s)) for _, s, k in self.specs if k in TRAINABLE_KINDS)
