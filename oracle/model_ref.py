"""Dense CPU oracle for the Lisec network -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (lisec_amd/) never does.

Op-for-op restatement, on dense rank-6 tensors with torch CPU ops, of

    RepeatLayer / MaxPoolingVFELayer        /root/reference/model_training.py:32-61
    addVFELayer / addFCN / addDenseLayer    model_training.py:155-186
    addConv3DLayer                          model_training.py:191-196
    addConv2DLayer / addRPNConvLayer        model_training.py:201-215
    createModel                             model_training.py:222-257
    compile(SGD nesterov, ['mse','mse'])    model_training.py:295-296
    predict (inference-mode BN)             Predict.py:38

PARITY UNPINNED: the arithmetic of these statements lives in TensorFlow/Keras, a
third-party dependency that is absent from /root/reference, not installed here and not
version-pinned by the reference (README.md:5-10 only names it); the reference ships no
tests, golden outputs or usable weights for this path (SampleModel/15SampleEpoch0.h5 is
stripped, .MISSING_LARGE_BLOBS:1).  What follows restates the PUBLISHED Keras semantics:
  * Dense(use_bias=False): x @ kernel, kernel (in, out);
  * BatchNormalization(): axis -1, epsilon 1e-3, momentum 0.99; training: batch mean and
    BIASED batch variance over every other axis; x*inv + (beta - mean*inv) with
    inv = gamma*rsqrt(var+eps); moving stats m <- m*0.99 + batch*0.01 (variance biased on
    the rank-6 non-fused path, Bessel-corrected on the fused rank-4/5 path);
  * Conv3D/Conv2D 'valid' after explicit ZeroPadding, cross-correlation, kernels
    (kd,kh,kw,in,out)/(kh,kw,in,out), bias;
  * Conv2DTranspose(padding='same'): out = in*stride, kernel (kh,kw,out,in);
  * reduce_max gradient: split equally among ties; relu gradient: (z > 0);
  * 'mse' = mean over every element of (pred - target)^2, the two losses summed;
  * SGD(lr, decay, momentum, nesterov): lr_t = lr/(1+decay*iter); v <- m*v - lr_t*g;
    w <- w + m*v - lr_t*g.
The only executable pins are shape known-answers: model.png from conv3d onward and
rpnToRegion.py:116-117 ((100,200,2) / (100,200,14)).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99

# (name, kind, meta) in forward order; kinds: dense, bn, conv3d, conv2d, deconv2d
RPN_BLOCKS = ((128, 3), (128, 5), (256, 5))       # (cout, q)  model_training.py:245-251
DECONVS = ((3, 1), (2, 2), (4, 4))                 # (kernel, stride) :246,:249,:252
MID = (((2, 1, 1), (1, 1, 1)), ((1, 1, 1), (0, 1, 1)), ((2, 1, 1), (1, 1, 1)))  # (stride, pad) :236-238


def param_specs(dprime=1):
    """[(name, shape, kind)] of every variable, in forward order (trainable and BN moving stats).  dprime: the depth the
    three Conv3D layers leave (1 for Constants.nz = 8); Permute + Reshape (model_training.py:242-243) folds it into the
    channels, so the first RPN conv reads 64 * dprime of them."""
    specs = []

    def bn(prefix, c):
        specs.extend([(prefix + ".gamma", (c,), "gamma"), (prefix + ".beta", (c,), "beta"),
                      (prefix + ".moving_mean", (c,), "moving_mean"),
                      (prefix + ".moving_variance", (c,), "moving_variance")])

    for name, cin, cout in (("vfe1", 6, 16), ("vfe2", 32, 32), ("fcn", 64, 64)):
        specs.append((f"{name}.dense.kernel", (cin, cout), "kernel"))
        bn(f"{name}.bn", cout)
    for i in range(3):
        specs.append((f"mid{i+1}.conv.kernel", (3, 3, 3, 64, 64), "kernel"))
        specs.append((f"mid{i+1}.conv.bias", (64,), "bias"))
        bn(f"mid{i+1}.bn", 64)
        specs.append((f"mid{i+1}.dense.kernel", (64, 64), "kernel"))
    cin = 64 * dprime
    for b, (cout, q) in enumerate(RPN_BLOCKS):
        for j in range(q + 1):
            specs.append((f"rpn{b+1}.conv{j}.kernel", (3, 3, cin, cout), "kernel"))
            specs.append((f"rpn{b+1}.conv{j}.bias", (cout,), "bias"))
            bn(f"rpn{b+1}.bn{j}", cout)
            cin = cout
        k, _ = DECONVS[b]
        specs.append((f"up{b+1}.kernel", (k, k, 256, cout), "kernel"))   # (kh,kw,out,in)
        specs.append((f"up{b+1}.bias", (256,), "bias"))
    specs.append(("cls.kernel", (1, 1, 768, 2), "kernel"))
    specs.append(("cls.bias", (2,), "bias"))
    specs.append(("reg.kernel", (1, 1, 768, 14), "kernel"))
    specs.append(("reg.bias", (14,), "bias"))
    return specs


def is_trainable(kind):
    return kind in ("kernel", "bias", "gamma", "beta")


def _bn(x, p, prefix, training, stats, fused):
    """BatchNormalization over the last axis; records batch stats in `stats` when training."""
    gamma, beta = p[prefix + ".gamma"], p[prefix + ".beta"]
    if training:
        dims = tuple(range(x.dim() - 1))
        mean = x.mean(dim=dims)
        var = ((x - mean) ** 2).mean(dim=dims)            # biased
        n = x.numel() // x.shape[-1]
        if stats is not None:
            stats[prefix] = (mean.detach(), var.detach(), n, fused)
    else:
        mean, var = p[prefix + ".moving_mean"], p[prefix + ".moving_variance"]
    inv = gamma * torch.rsqrt(var + BN_EPS)
    return x * inv + (beta - mean * inv)


def _fcn(x, p, name, training, stats):
    y = x @ p[f"{name}.dense.kernel"]                     # addDenseLayer, no bias (:184)
    y = _bn(y, p, f"{name}.bn", training, stats, fused=False)
    return F.relu(y)                                       # :173


def _vfe(x, p, name, training, stats):
    a = _fcn(x, p, name, training, stats)                  # :158
    pooled = a.max(dim=-2, keepdim=True).values            # MaxPoolingVFELayer (:56) -- pad rows take part
    rep = pooled.expand(*a.shape[:-1], pooled.shape[-1])   # RepeatLayer (:40)
    return torch.cat([rep, a], dim=-1)                     # pooled half first (:164-165)


def forward(params, x, training=False, stats=None, taps=None):
    """params: dict name -> torch tensor; x: (B, D, H, W, T, 6).  Returns (cls, reg).

    taps (optional dict) receives intermediate activations for layer-by-layer parity tests.
    """
    p = params
    h = _vfe(x, p, "vfe1", training, stats)                # :231
    h = _vfe(h, p, "vfe2", training, stats)                # :232
    h = _fcn(h, p, "fcn", training, stats)                 # :233
    h = h.max(dim=-2).values                               # MaxPoolingVFELayer(combine=True) :235
    if taps is not None:
        taps["vfe_grid"] = h
    return forward_from_grid(p, h, training, stats, taps)


def forward_from_grid(params, h, training=False, stats=None, taps=None):
    """The graph from the first Conv3D on (model_training.py:236-255); h: (B, D, H, W, 64), the output of
    MaxPoolingVFELayer(combine=True).  Lets the full Lyft grid be checked on a CPU: the VFE part comes from the
    sparse-exact oracle (oracle/vfe_sparse_ref.py, proven equal to the dense VFE), the rest is dense."""
    p = params
    for i, (stride, pad) in enumerate(MID):                # :236-238
        w = p[f"mid{i+1}.conv.kernel"].permute(4, 3, 0, 1, 2)
        y = F.conv3d(h.permute(0, 4, 1, 2, 3), w, p[f"mid{i+1}.conv.bias"], stride=stride, padding=pad)
        y = y.permute(0, 2, 3, 4, 1)
        y = _bn(y, p, f"mid{i+1}.bn", training, stats, fused=True)
        h = F.relu(y @ p[f"mid{i+1}.dense.kernel"])        # addDenseLayer(64, 'relu') :195
        if taps is not None:
            taps[f"mid{i+1}"] = h
    B, D, H, W, C = h.shape
    h = h.permute(0, 2, 3, 4, 1).reshape(B, H, W, C * D)   # Permute((2,3,4,1)) + Reshape :242-243
    ups = []
    for b, (cout, q) in enumerate(RPN_BLOCKS):             # :245-252
        for j in range(q + 1):
            w = p[f"rpn{b+1}.conv{j}.kernel"].permute(3, 2, 0, 1)
            y = F.conv2d(h.permute(0, 3, 1, 2), w, p[f"rpn{b+1}.conv{j}.bias"],
                         stride=2 if j == 0 else 1, padding=1)
            y = _bn(y.permute(0, 2, 3, 1), p, f"rpn{b+1}.bn{j}", training, stats, fused=True)
            if taps is not None:
                taps[f"rpn{b+1}.z{j}"] = y                      # pre-ReLU (kink diagnostics in tests)
            h = F.relu(y)
        if taps is not None:
            taps[f"rpn{b+1}"] = h
        k, s = DECONVS[b]
        wt = p[f"up{b+1}.kernel"].permute(3, 2, 0, 1)      # (kh,kw,out,in) -> (in,out,kh,kw)
        u = F.conv_transpose2d(h.permute(0, 3, 1, 2), wt, p[f"up{b+1}.bias"], stride=s,
                               padding=(k - s) // 2)       # padding='same' -> out = in*stride
        ups.append(u.permute(0, 2, 3, 1))
    cat = torch.cat(ups, dim=-1)                           # :253
    if taps is not None:
        taps["concat"] = cat
    cls = cat @ p["cls.kernel"][0, 0] + p["cls.bias"]      # 1x1 conv, linear (:254)
    reg = cat @ p["reg.kernel"][0, 0] + p["reg.bias"]      # :255
    return cls, reg


def mse_loss(cls, reg, y_cls, y_reg):
    """compile(loss=['mse','mse']) (model_training.py:296): sum of the two per-output means."""
    return ((cls - y_cls) ** 2).mean() + ((reg - y_reg) ** 2).mean()


def updated_moving_stats(params, stats):
    """Keras moving-average update after one training step."""
    out = {}
    for prefix, (mean, var, n, fused) in stats.items():
        v = var * (n / max(n - 1, 1)) if fused else var
        out[prefix + ".moving_mean"] = params[prefix + ".moving_mean"] * BN_MOMENTUM + mean * (1 - BN_MOMENTUM)
        out[prefix + ".moving_variance"] = params[prefix + ".moving_variance"] * BN_MOMENTUM + v * (1 - BN_MOMENTUM)
    return out


def sgd_nesterov_step(w, g, v, iteration, lr=0.01, decay=1e-6, momentum=0.9):
    """optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True) (model_training.py:295)."""
    lr_t = lr / (1.0 + decay * iteration)
    v_new = momentum * v - lr_t * g
    w_new = w + momentum * v_new - lr_t * g
    return w_new, v_new


def train_step(params, velocity, x, y_cls, y_reg, iteration):
    """One fit() step with batch_size=1 semantics (model_training.py:299).

    params: dict of torch tensors (trainables must NOT require grad; handled here).
    Returns (loss, grads dict, new params dict, new velocity dict).
    """
    specs = param_specs()
    work = {}
    for name, _, kind in specs:
        t = params[name].detach().clone()
        if is_trainable(kind):
            t.requires_grad_(True)
        work[name] = t
    stats = {}
    cls, reg = forward(work, x, training=True, stats=stats)
    loss = mse_loss(cls, reg, y_cls, y_reg)
    loss.backward()
    grads, new_p, new_v = {}, {}, {}
    for name, _, kind in specs:
        if is_trainable(kind):
            g = work[name].grad if work[name].grad is not None else torch.zeros_like(work[name])
            grads[name] = g
            w, v = sgd_nesterov_step(params[name], g, velocity[name], iteration)
            new_p[name], new_v[name] = w.detach(), v.detach()
        else:
            new_p[name] = params[name]
    new_p.update(updated_moving_stats(params, stats))
    return loss.detach(), grads, new_p, new_v, (cls.detach(), reg.detach())


def glorot_params(seed=1234, dtype=torch.float32, randomize_bn=False, dprime=1):
    """Keras default initialisers (glorot_uniform kernels, zero biases, BN gamma=1/beta=0/mean=0/var=1).
    randomize_bn=True draws non-trivial BN variables so parity tests exercise them."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape, kind in param_specs(dprime):
        if kind == "kernel":
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            limit = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
            a = rng.uniform(-limit, limit, shape)
        elif kind in ("bias", "beta", "moving_mean"):
            a = rng.normal(0, 0.1, shape) if randomize_bn else np.zeros(shape)
        elif kind == "gamma":
            a = rng.uniform(0.5, 1.5, shape) * rng.choice([1.0, 1.0, 1.0, -1.0], shape) if randomize_bn \
                else np.ones(shape)
        else:  # moving_variance
            a = rng.uniform(0.5, 1.5, shape) if randomize_bn else np.ones(shape)
        out[name] = torch.tensor(a, dtype=dtype)
    return out
