"""End-to-end forward (voxeliser -> VFE -> 3D-conv middle -> RPN heads) through the C ABI vs the
dense oracle (oracle/model_ref.py) on identical voxel grids.
Tolerance: BASELINE north_star -- RPN outputs rtol 1e-3, paired with atol 1e-3*max|ref|."""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=8, maxVoxelY=16, maxVoxelZ=8)


def small_cloud(seed=1, n=3000):
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 2.1, n)], 1)
    pts[:600, :2] *= 0.1
    pts[:600, 2] = 0.5 + 0.5 * rng.uniform(0, 1, 600)
    return pts.astype(np.float32)


def close(got, ref, rtol=1e-3, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    atol = rtol * np.abs(ref).max()
    err = np.abs(got - ref)
    assert (err <= atol + rtol * np.abs(ref)).all(), f"{what}: max err {err.max():.3e}, atol {atol:.3e}"
    return err.max() / max(np.abs(ref).max(), 1e-30)


def loose_names(taps, names, eps=5e-6):
    """Parameters whose gradient a ReLU 'kink' can disturb.  A pre-activation of RPN layer (b, j) within fp32 noise of 0
    (|z| < eps in the fp64 oracle) flips one gradient gate between the oracle and the fp32 GPU run; that reaches the
    gradients of that layer and of every layer in FRONT of it, never the ones behind it.  So only the parameters up to and
    including the last kinked layer (in forward order) are compared at the loose bound; everything behind keeps the tight
    one (a kink anywhere used to loosen every tensor to 10 %)."""
    last = None
    for k, v in taps.items():
        m = re.fullmatch(r"rpn(\d)\.z(\d)", k)
        if m and float(v.abs().min()) < eps:
            key = (int(m.group(1)), int(m.group(2)))
            last = key if last is None or key > last else last
    if last is None:
        return set()
    cut = max(i for i, n in enumerate(names) if n.startswith(f"rpn{last[0]}.bn{last[1]}."))
    return set(names[:cut + 1])


def oracle_forward(op, pts, cfg, training, dtype=torch.float64):
    from oracle import model_ref as M
    from oracle import voxel_ref
    vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **cfg)
    shape = (cfg["maxVoxelZ"], 2 * cfg["maxVoxelX"], 2 * cfg["maxVoxelY"], cfg["sampleSize"], 6)
    dense = torch.from_numpy(voxel_ref.to_dense(vox, shape))[None].to(dtype)
    p = {k: v.to(dtype) for k, v in op.items()}
    taps, stats = {}, {}
    cls, reg = M.forward(p, dense, training=training, stats=stats, taps=taps)
    return cls, reg, taps, stats, p


@pytest.mark.parametrize("compose_head", [True, False])
@pytest.mark.parametrize("training", [False, True])
def test_forward_small_grid_vs_dense_oracle(training, compose_head):
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M

    op = M.glorot_params(seed=21, randomize_bn=True)
    pts = small_cloud()
    cls_r, reg_r, taps, stats, p64 = oracle_forward(op, pts, SMALL, training)
    dev = torch.device("cuda")
    net = LisecNet(16, 32, 8, 35, params=ParamStore(dev, init=op), compose_head=compose_head)
    sample = Voxelizer(**SMALL)(pts)
    cls, reg = net.forward(sample, training=training)
    torch.cuda.synchronize()
    a = net.act
    close(net.dense_grid().cpu().numpy(), taps["vfe_grid"][0].numpy(), what="vfe grid")
    for i in (1, 2, 3):
        close(a[f"mid{i}.u"].cpu().numpy(), taps[f"mid{i}"][0].numpy(), what=f"mid{i}")
    if not compose_head:                 # the collapsed form never builds the (Ho,Wo,768) concat
        close(a["concat"].cpu().numpy(), taps["concat"][0].numpy(), what="concat")
    else:
        assert "concat" not in a
    close(cls.cpu().numpy(), cls_r.numpy(), what="cls")
    close(reg.cpu().numpy(), reg_r.numpy(), what="reg")
    assert cls.shape == (1, 8, 16, 2) and reg.shape == (1, 8, 16, 14)
    if training:
        new = M.updated_moving_stats(p64, stats)
        got = net.params.to_dict()
        for k, v in new.items():
            close(got[k], v.numpy(), rtol=1e-4, what=k)


@pytest.mark.parametrize("loss", ["mse", "smoothl1_ce"])
def test_train_steps_small_grid_vs_oracle_autograd(loss):
    """Three fit() steps (forward with batch statistics, backward, SGD-Nesterov) vs the dense oracle's
    torch-autograd step in fp64: loss, every gradient, every updated variable, velocity carried over.

    Before each step the GPU state is re-synchronised with the oracle's: on this toy grid (8 RPN
    positions in block 3) the fp32 and fp64 TRAJECTORIES diverge chaotically within two steps -- the
    oracle run in fp32 vs fp64 shows the same (tools/debug_train.py) -- so only per-step quantities are
    comparable.  A pre-ReLU value within fp32 noise of 0 flips a ReLU gradient; such steps (detected
    on the oracle's side) are compared at a loose tolerance."""
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M
    from oracle import voxel_ref
    import torch.nn.functional as F

    op = M.glorot_params(seed=33, randomize_bn=True)
    dev = torch.device("cuda")
    net = LisecNet(16, 32, 8, 35, params=ParamStore(dev, init=op))
    net._prepare_training()
    vox = Voxelizer(**SMALL)
    rng = np.random.default_rng(8)
    p64 = {k: v.double() for k, v in op.items()}
    vel = {n: torch.zeros_like(p64[n]) for n, _, k in M.param_specs() if M.is_trainable(k)}
    shape = (8, 16, 32, 35, 6)
    for it in range(3):
        p64 = {k: v.float().double() for k, v in p64.items()}
        vel = {k: v.float().double() for k, v in vel.items()}
        net.params.load_dict({k: v.float() for k, v in p64.items()})
        for n_, v_ in vel.items():
            net.params.grad_view(net.velocity, n_).copy_(v_.float())
        net.iterations = it
        pts = small_cloud(seed=40 + it)
        y_cls = rng.integers(0, 3, (8, 16, 2)).astype(np.float32)
        y_reg = rng.normal(0, 1, (8, 16, 14)).astype(np.float32)
        ref_vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **SMALL)
        dense = torch.from_numpy(voxel_ref.to_dense(ref_vox, shape))[None].double()
        yc, yr = torch.from_numpy(y_cls)[None].double(), torch.from_numpy(y_reg)[None].double()
        taps = {}
        M.forward(p64, dense, training=True, stats={}, taps=taps)
        order = [n for n, _, k in M.param_specs() if M.is_trainable(k)]
        loose = loose_names(taps, order)
        if loss == "mse":
            loss_r, grads_r, p64_new, vel_new, _ = M.train_step(p64, vel, dense, yc, yr, it)
        else:
            work = {n: p64[n].clone().requires_grad_(M.is_trainable(k)) for n, _, k in M.param_specs()}
            cls, reg = M.forward(work, dense, training=True, stats={})
            loss_r = F.binary_cross_entropy_with_logits(cls, yc.clamp(0, 1)) + F.smooth_l1_loss(reg, yr)
            loss_r.backward()
            grads_r = {n: work[n].grad for n, _, k in M.param_specs() if M.is_trainable(k)}
            p64_new, vel_new = None, None
            loss_r = loss_r.detach()
        sample = vox(pts)
        lo = net.train_step(sample, torch.from_numpy(y_cls).to(dev), torch.from_numpy(y_reg).to(dev), loss=loss)
        torch.cuda.synchronize()
        assert abs(lo[0].item() - loss_r.item()) <= 1e-5 * abs(loss_r.item())
        for name, g in grads_r.items():
            got = net.params.grad_view(net.grad, name).cpu().numpy()
            ref = g.numpy()
            if ".conv" in name and name.endswith(".bias") and np.abs(ref).max() < 1e-12:
                # bias feeding a training-mode BN: the exact gradient is 0; ours is rounding noise
                assert np.abs(got).max() < 1e-5, name
                continue
            gtol = 1e-1 if name in loose else 3e-3
            tol = gtol * np.abs(ref).max() + 1e-7
            err = np.abs(got - ref).max()
            assert err <= tol, f"step {it} grad {name}: err {err:.3e} tol {tol:.3e} (max ref {np.abs(ref).max():.3e})"
        if p64_new is None:
            return
        got_p = net.params.to_dict()
        for name, v in p64_new.items():
            close(got_p[name], v.numpy(), rtol=(1e-2 if name in loose else 1e-4), what=f"step {it} param {name}")
        for n_, v_ in vel_new.items():
            gv = net.params.grad_view(net.velocity, n_).cpu().numpy()
            if np.abs(v_.numpy()).max() < 1e-12:          # velocity of an exactly-zero gradient (see above)
                assert np.abs(gv).max() < 1e-6, n_
                continue
            close(gv, v_.numpy(), rtol=(1e-1 if n_ in loose else 3e-3), what=f"step {it} velocity {n_}")
        p64, vel = p64_new, vel_new


def test_train_step_fully_occupied_grid_vs_oracle():
    """Every cell of the grid occupied (a dense (D,H,W,T,6) array handed to the model, as Model.fit accepts): there is
    no "empty cell" class, the constant row V of the per-voxel outputs is defined as 0 and must not be read from
    uninitialised memory (the saved buffer is poisoned with NaN first).  One training step vs the dense
    autograd oracle in fp64."""
    from lisec_amd import _lib
    from lisec_amd import model_training as mt
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from oracle import model_ref as M

    rng = np.random.default_rng(12)
    D, H, W, T = 8, 16, 32, 35
    dense = rng.normal(0, 1, (D, H, W, T, 6)).astype(np.float32)
    dense[:, :, :, 20:, :] *= (rng.uniform(0, 1, (D, H, W, 1, 1)) < 0.5)       # some voxels with zero (pad-like) rows
    assert (np.abs(dense).reshape(D * H * W, -1).max(1) > 0).all()
    op = M.glorot_params(seed=34, randomize_bn=True)
    dev = torch.device("cuda")
    net = LisecNet(H, W, D, T, params=ParamStore(dev, init=op))
    sample = mt.dense_to_sample(dense, dev)
    assert sample.host_info()["V"] == D * H * W
    net.vfe._saved = torch.full((_lib.load().lisec_vfe_saved_floats_rows(sample.cap, sample.n_points),), float("nan"),
                                device=dev)
    y_cls = rng.integers(0, 3, (H // 2, W // 2, 2)).astype(np.float32)
    y_reg = rng.normal(0, 1, (H // 2, W // 2, 14)).astype(np.float32)
    p64 = {k: v.double() for k, v in op.items()}
    vel = {n: torch.zeros_like(p64[n]) for n, _, k in M.param_specs() if M.is_trainable(k)}
    x = torch.from_numpy(dense)[None].double()
    taps = {}
    cls_r, reg_r = M.forward(p64, x, training=True, stats={}, taps=taps)
    loose = loose_names(taps, [n for n, _, k in M.param_specs() if M.is_trainable(k)])
    loss_r, grads_r, _, _, _ = M.train_step(p64, vel, x, torch.from_numpy(y_cls)[None].double(),
                                            torch.from_numpy(y_reg)[None].double(), 0)
    lo = net.train_step(sample, torch.from_numpy(y_cls).to(dev), torch.from_numpy(y_reg).to(dev))
    torch.cuda.synchronize()
    close(net.dense_grid().cpu().numpy(), taps["vfe_grid"][0].numpy(), what="vfe grid (all cells occupied)")
    assert torch.isfinite(net.grad).all() and torch.isfinite(net.params.theta).all()
    vout = net.vfe.saved_field("vout").cpu().numpy()
    assert (vout[D * H * W] == 0).all() and (net.vfe.saved_field("delta").cpu().numpy()[D * H * W] == 0).all()
    assert abs(lo[0].item() - loss_r.item()) <= 1e-5 * abs(loss_r.item())
    for name, g in grads_r.items():
        got = net.params.grad_view(net.grad, name).cpu().numpy()
        ref = g.numpy()
        if ".conv" in name and name.endswith(".bias") and np.abs(ref).max() < 1e-12:
            assert np.abs(got).max() < 1e-5, name
            continue
        err, tol = np.abs(got - ref).max(), (1e-1 if name in loose else 3e-3) * np.abs(ref).max() + 1e-7
        assert err <= tol, f"grad {name}: err {err:.3e} tol {tol:.3e}"


def test_depth_fold_nz12_forward_and_training_step(tmp_path):
    """Permute((2,3,4,1)) + Reshape in its general form (model_training.py:242-243): nz = 12 leaves a depth of 2 after
    the three Conv3D layers, the RPN then reads 128 = 64*2 channels with channel index c*2 + d.  (Constants.nz = 8 folds
    to depth 1, where the pair is a view.)  Forward in both BatchNormalization modes and one training step vs the dense
    oracle in fp64."""
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore, fold_depth
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M
    from oracle import voxel_ref

    assert fold_depth(8) == 1 and fold_depth(12) == 2 and fold_depth(16) == 3
    cfg = dict(SMALL, maxVoxelZ=12)
    rng = np.random.default_rng(21)
    n = 3500
    pts = np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 3.1, n)], 1).astype(np.float32)
    op = M.glorot_params(seed=55, randomize_bn=True, dprime=2)
    assert tuple(op["rpn1.conv0.kernel"].shape) == (3, 3, 128, 128)
    dev = torch.device("cuda")
    net = LisecNet(16, 32, 12, 35, params=ParamStore(dev, init=op))
    assert net.dprime == 2 and net.act["fold"].shape == (16, 32, 128)
    sample = Voxelizer(**cfg)(pts)
    ref_vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **cfg)
    dense = torch.from_numpy(voxel_ref.to_dense(ref_vox, (12, 16, 32, 35, 6)))[None].double()
    p64 = {k: v.double() for k, v in op.items()}
    for training in (False, True):
        cls, reg = net.forward(sample, training=training)
        taps = {}
        cls_r, reg_r = M.forward(p64, dense, training=training, stats={}, taps=taps)
        close(cls.cpu().numpy(), cls_r.numpy(), what=f"class map (nz=12, training={training})")
        close(reg.cpu().numpy(), reg_r.numpy(), what=f"regression map (nz=12, training={training})")
    loose = loose_names(taps, [n for n, _, k in M.param_specs(2) if M.is_trainable(k)])
    y_cls = rng.integers(0, 3, (8, 16, 2)).astype(np.float32)
    y_reg = rng.normal(0, 1, (8, 16, 14)).astype(np.float32)
    vel = {n_: torch.zeros_like(p64[n_]) for n_, _, k in M.param_specs(2) if M.is_trainable(k)}
    loss_r, grads_r, _, _, _ = M.train_step(p64, vel, dense, torch.from_numpy(y_cls)[None].double(),
                                            torch.from_numpy(y_reg)[None].double(), 0)
    lo = net.train_step(sample, torch.from_numpy(y_cls).to(dev), torch.from_numpy(y_reg).to(dev))
    torch.cuda.synchronize()
    assert abs(lo[0].item() - loss_r.item()) <= 1e-5 * abs(loss_r.item())
    for name, g in grads_r.items():
        got = net.params.grad_view(net.grad, name).cpu().numpy()
        ref = g.numpy()
        if ".conv" in name and name.endswith(".bias") and np.abs(ref).max() < 1e-12:
            assert np.abs(got).max() < 1e-5, name
            continue
        err, tol = np.abs(got - ref).max(), (1e-1 if name in loose else 3e-3) * np.abs(ref).max() + 1e-7
        assert err <= tol, f"grad {name}: err {err:.3e} tol {tol:.3e}"
    # the drop-in surface with this nz: createModel -> predict -> save (Keras-layout .h5 and .npz) -> load_model
    from lisec_amd import model_training as mt
    model = mt.createModel(16, 32, 12, 35)
    want = model.predict(mt.SparseVoxels(sample))
    for name in ("m12.h5", "m12.npz"):
        model.save(str(tmp_path / name))
        again = mt.load_model(str(tmp_path / name))
        assert again.nz == 12 and again.net.dprime == 2
        got = again.predict(mt.SparseVoxels(sample))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def _oracle_loss(cls, reg, yc, yr, loss):
    """'mse': compile(loss=['mse','mse']) (model_training.py:296); 'smoothl1_ce': the sigmoid cross-entropy +
    SmoothL1 pair BASELINE.json's config 4 names (lisec_rpn_loss kind 1)."""
    from oracle import model_ref as M
    import torch.nn.functional as F
    if loss == "mse":
        return M.mse_loss(cls, reg, yc, yr)
    return F.binary_cross_entropy_with_logits(cls, yc.clamp(0, 1)) + F.smooth_l1_loss(reg, yr)


def _hybrid_oracle_lyft(op, pts, training, y_cls=None, y_reg=None, dtype=torch.float64, loss="mse"):
    """Full Lyft grid on the CPU: VFE by the fp64 row-class oracle (proven equal to the dense VFE in
    tests/test_oracle_model.py), everything from the first Conv3D on by the dense torch oracle in fp64
    (autograd for the gradients), VFE gradients by the row-class backward."""
    from conftest import LYFT
    from oracle import model_ref as M
    from oracle import vfe_sparse_ref as S
    from oracle import voxel_ref
    D, H, W, T = 8, 200, 400, 35
    ncells = D * H * W
    vox = voxel_ref.voxelize_ref(pts.astype(np.float64), **LYFT)
    x, w, vid, seg = S.build_rows(vox["feats"], vox["npts"], T, ncells)
    pn = {k: v.double().numpy() for k, v in op.items()}
    out, cache = S.forward(pn, x, w, vid, seg, N=float(ncells * T), training=training)
    c = vox["coords"]
    cells = (c[:, 0] * H + c[:, 1]) * W + c[:, 2]
    grid = np.empty((ncells, 64))
    grid[:] = out[-1]
    grid[cells] = out[:-1]
    p64 = {k: v.to(dtype) for k, v in op.items()}
    g = torch.from_numpy(grid.reshape(1, D, H, W, 64)).to(dtype)
    if not training:
        with torch.no_grad():
            cls, reg = M.forward_from_grid(p64, g, training=False)
        return cls, reg, None, None
    names = [n for n, _, k in M.param_specs() if M.is_trainable(k) and n.split(".")[0] not in ("vfe1", "vfe2", "fcn")]
    work = dict(p64)
    for n in names:
        work[n] = p64[n].clone().requires_grad_(True)
    g.requires_grad_(True)
    cls, reg = M.forward_from_grid(work, g, training=True, stats={})
    loss = _oracle_loss(cls, reg, torch.from_numpy(y_cls)[None].to(dtype), torch.from_numpy(y_reg)[None].to(dtype), loss)
    loss.backward()
    grads = {n: work[n].grad.double().numpy() for n in names}
    dg = g.grad.double().numpy().reshape(ncells, 64)
    empty = np.ones(ncells, bool)
    empty[cells] = False
    dout = np.concatenate([dg[cells], dg[empty].sum(0, keepdims=True)])
    grads.update(S.backward(pn, cache, dout))
    return cls.detach(), reg.detach(), loss.item(), grads


def u20k(seed, n=20000):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-55, 55, n), rng.uniform(-55, 55, n), rng.uniform(-0.5, 2.5, n)], 1).astype(np.float32)


def full_grid_gradient_report(pts, loss, seed=5, wseed=77):
    """One training step at the Lyft grid on the GPU and in the CPU oracle (fp64 = truth, fp32 = what the SAME math
    gives in the product's arithmetic).  Returns (maps/loss dict, rows) with rows = (name, max|ref|, gpu relative
    error, fp32-oracle relative error), both relative to max|ref| of that tensor; out["l2"][name] = relative L2
    distances (gpu vs fp64, fp32 oracle vs fp64, gpu vs fp32 oracle)."""
    from conftest import LYFT
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M
    rng = np.random.default_rng(seed)
    op = M.glorot_params(seed=wseed, randomize_bn=True)
    dev = torch.device("cuda")
    net = LisecNet(200, 400, 8, 35, params=ParamStore(dev, init=op))
    sample = Voxelizer(**LYFT)(pts)
    y_cls = rng.integers(0, 3, (100, 200, 2)).astype(np.float32)
    y_reg = rng.normal(0, 1, (100, 200, 14)).astype(np.float32)
    net.forward(sample, training=True)
    lo = net.backward(torch.from_numpy(y_cls).to(dev), torch.from_numpy(y_reg).to(dev), loss=loss)
    torch.cuda.synchronize()
    cls_t, reg_t, loss_r, grads_r = _hybrid_oracle_lyft(op, pts, True, y_cls, y_reg, loss=loss)
    _, _, _, grads_32 = _hybrid_oracle_lyft(op, pts, True, y_cls, y_reg, dtype=torch.float32, loss=loss)
    rows, l2 = [], {}
    for name, ref in grads_r.items():
        got = net.params.grad_view(net.grad, name).cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max()
        nrm = max(float(np.linalg.norm(ref)), 1e-300)
        l2[name] = (float(np.linalg.norm(got - ref)) / nrm, float(np.linalg.norm(grads_32[name] - ref)) / nrm,
                    float(np.linalg.norm(got - grads_32[name])) / nrm)
        if ".conv" in name and name.endswith(".bias") and scale < 1e-10:
            # bias of a conv feeding a training-mode BatchNormalization: the exact gradient is 0
            assert np.abs(got).max() < 1e-5, name
            continue
        rows.append((name, scale, np.abs(got - ref).max() / scale, np.abs(grads_32[name] - ref).max() / scale))
    out = dict(head=net.act["head"].cpu().numpy(), cls=cls_t[0].numpy(), reg=reg_t[0].numpy(), loss=float(lo[0].item()),
               loss_ref=loss_r, l2=l2)
    return out, rows


def test_full_lyft_grid_inference_vs_oracle():
    """BASELINE config 3 at the real grid (8,200,400,35): RPN class/regression maps in inference mode vs the CPU
    oracle in fp64, rtol 1e-3."""
    from conftest import LYFT
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M
    pts = u20k(5)
    op = M.glorot_params(seed=77, randomize_bn=True)
    net = LisecNet(200, 400, 8, 35, params=ParamStore(torch.device("cuda"), init=op))
    cls, reg = net.forward(Voxelizer(**LYFT)(pts), training=False)
    cls_r, reg_r, _, _ = _hybrid_oracle_lyft(op, pts, training=False)
    close(cls.cpu().numpy(), cls_r.numpy(), what="class map (inference)")
    close(reg.cpu().numpy(), reg_r.numpy(), what="regression map (inference)")


# Per-tensor gradient bound at the full grid.  What was measured (tools/grad_conditioning.py, profiles/r02_grad_conditioning.txt):
#   * every gradient below the last BatchNormalization of an RPN block carries a relative L2 error of 0.5-1 % against
#     the fp64 oracle -- in the fp32 ORACLE (torch CPU, same math) exactly as on the GPU, and the two fp32 results differ
#     from EACH OTHER by as much (~0.75 %): it is the noise floor of this network in fp32, not an implementation error.
#     Source: the pre-BN maps are stored in fp32 and 98 % of the grid holds one constant, so a channel's spread over the
#     positions is 1e-3..1e-5 of its mean and (y - mean)/std amplifies the 6e-8 rounding of y by that factor.  Evaluating
#     the BatchNormalization backward itself in fp64 changes nothing (tried in the oracle); a denser sweep lowers it.
#   * two independent fp32 evaluations therefore cannot agree per tensor to a factor 1.5 of each other's error: the
#     measured per-tensor ratio gpu/fp32-oracle spreads over 0.6-2.4 (median 1.02) on the U20k sweep.
# Hence, per tensor in the relative L2 norm (a mis-scaled or mis-wired tensor shows as O(1) there, not as 1 %):
#   error(gpu) <= max(FLAT, SPREAD x error(fp32 oracle) on that SAME tensor), and over all tensors the MEDIAN ratio
#   must stay near 1 (a systematic loss of accuracy would move it), plus a max-norm sanity bound.
FLAT, SPREAD = 3e-3, 3.0


def _check_gradients(out, rows, median_bound):
    ratios, bad = [], []
    for name, scale, e_max, o_max in rows:
        e, o, _ = out["l2"][name]
        if e > max(FLAT, SPREAD * o):
            bad.append(f"{name}: L2 gpu {e:.2e} vs fp32-oracle {o:.2e} (max-norm {e_max:.2e} vs {o_max:.2e})")
        if o > 1e-4:
            ratios.append(e / o)
    assert not bad, "gradients beyond max(3e-3, 3 x the fp32 oracle's own error on that tensor):\n" + "\n".join(bad)
    med = float(np.median(ratios))
    assert med <= median_bound, f"median gpu/fp32-oracle error ratio {med:.2f} > {median_bound}"
    assert max(r[2] for r in rows) < 0.1
    return med


@pytest.mark.parametrize("loss", ["mse", "smoothl1_ce"])
def test_full_lyft_grid_training_step_vs_oracle(loss):
    """BASELINE config 4 at the real grid, U20k sweep: training-mode maps, the loss and EVERY gradient of one step,
    for the reference's MSE+MSE pair (model_training.py:296) and for SmoothL1 + cross-entropy (config 4 as worded)."""
    out, rows = full_grid_gradient_report(u20k(5), loss)
    close(out["head"][:, :, :2], out["cls"], what="class map (training)")
    close(out["head"][:, :, 2:], out["reg"], what="regression map (training)")
    assert abs(out["loss"] - out["loss_ref"]) <= 1e-5 * abs(out["loss_ref"])
    _check_gradients(out, rows, median_bound=1.3)


def dense_sweep(seed, n=700000):
    """A sweep that fills two thirds of the 640 000 cells: the maps vary from position to position and the fp32 noise
    floor of the oracle drops from ~0.75 % to ~0.45 % (the GPU's stays at ~0.7 %: 1.6x, measured)."""
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-49.4, 49.9, n), rng.uniform(-49.7, 49.9, n), rng.uniform(0.26, 1.99, n)],
                    1).astype(np.float32)


def test_full_lyft_grid_training_step_dense_sweep():
    """The same step on a second full-grid case with a very different occupancy (two thirds of the cells non-empty,
    ~430 000 voxels): maps, loss and every gradient."""
    out, rows = full_grid_gradient_report(dense_sweep(6), "mse", seed=6)
    close(out["head"][:, :, :2], out["cls"], what="class map (training, dense sweep)")
    close(out["head"][:, :, 2:], out["reg"], what="regression map (training, dense sweep)")
    assert abs(out["loss"] - out["loss_ref"]) <= 1e-5 * abs(out["loss_ref"])
    _check_gradients(out, rows, median_bound=2.0)


def test_lyft_grid_r200k_cloud_and_empty_cloud():
    """A Lyft-size sweep (n = 200 000, crowded voxels well beyond 35 points, model_training.py:116) and the
    degenerate empty sweep: inference maps vs the hybrid oracle, and a training step that stays finite."""
    from conftest import LYFT
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M

    rng = np.random.default_rng(9)
    n = 200_000
    az = rng.uniform(0, 2 * np.pi, n)
    r = 2.0 + 68.0 * rng.uniform(0, 1, n) ** 2
    pts = np.stack([r * np.cos(az), r * np.sin(az), rng.uniform(-0.2, 2.2, n)], 1).astype(np.float32)
    op = M.glorot_params(seed=78, randomize_bn=True)
    dev = torch.device("cuda")
    net = LisecNet(200, 400, 8, 35, params=ParamStore(dev, init=op))
    vox = Voxelizer(**LYFT)
    sample = vox(pts)
    assert sample.host_info()["max_count"] > 35
    cls, reg = net.forward(sample, training=False)
    cls_r, reg_r, _, _ = _hybrid_oracle_lyft(op, pts, training=False)
    close(cls.cpu().numpy(), cls_r.numpy(), what="class map, R200k")
    close(reg.cpu().numpy(), reg_r.numpy(), what="regression map, R200k")
    y_cls = torch.zeros(100, 200, 2, device=dev)
    y_reg = torch.zeros(100, 200, 14, device=dev)
    lo = net.train_step(sample, y_cls, y_reg)
    torch.cuda.synchronize()
    assert torch.isfinite(lo).all() and torch.isfinite(net.grad).all() and torch.isfinite(net.params.theta).all()

    # empty sweep: every cell is "empty", the network output is position-class constant and finite
    net2 = LisecNet(200, 400, 8, 35, params=ParamStore(dev, init=op))
    empty = vox(np.zeros((0, 3), np.float32))
    cls0, reg0 = net2.forward(empty, training=False)
    cls0_r, reg0_r, _, _ = _hybrid_oracle_lyft(op, np.zeros((0, 3), np.float32), training=False)
    close(cls0.cpu().numpy(), cls0_r.numpy(), what="class map, empty sweep")
    close(reg0.cpu().numpy(), reg0_r.numpy(), what="regression map, empty sweep")
    lo = net2.train_step(empty, y_cls, y_reg)
    torch.cuda.synchronize()
    assert torch.isfinite(lo).all() and torch.isfinite(net2.grad).all()


def _field_case(name):
    from conftest import LYFT
    rng = np.random.default_rng(31)
    if name == "small":
        return SMALL, (16, 32), small_cloud(seed=4)
    if name == "small_edges":
        # voxels pressed against every face of the grid (boundary classes of all three axes) plus one crowded column
        n = 1500
        pts = np.stack([rng.choice([-3.9, -3.6, 3.6, 3.9], n) + rng.uniform(-0.05, 0.05, n),
                        rng.choice([-3.95, 0.0, 3.9], n) + rng.uniform(-0.05, 0.05, n),
                        rng.choice([0.3, 0.6, 1.8], n) + rng.uniform(-0.02, 0.02, n)], 1)
        return SMALL, (16, 32), pts.astype(np.float32)
    if name == "lyft_u20k":
        return LYFT, (200, 400), u20k(12)
    if name == "lyft_clustered":
        # 60 000 points in a few dense blobs: neighbouring voxels, i.e. many taps per output position
        c = rng.uniform(-30, 30, (40, 2))
        k = rng.integers(0, 40, 60000)
        pts = np.stack([c[k, 0] + rng.normal(0, 1.5, 60000), c[k, 1] + rng.normal(0, 1.5, 60000),
                        rng.uniform(0.26, 1.99, 60000)], 1)
        return LYFT, (200, 400), pts.astype(np.float32)
    if name == "lyft_empty":
        return LYFT, (200, 400), np.zeros((0, 3), np.float32)
    raise KeyError(name)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", ["small", "small_edges", "lyft_u20k", "lyft_clustered", "lyft_empty"])
def test_field_conv_equals_dense_contraction(case, training):
    """The first Conv3D evaluated over the VFE's compact output (constant + voxel rows, csrc/field_conv.hip) against
    the dense contraction over the materialised grid (model_training.py:236): pre-BN map, batch statistics, moving
    statistics and the network outputs.  Same math, different fp32 summation order: 2e-5 of the map's range."""
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M

    cfg, (nx, ny), pts = _field_case(case)
    op = M.glorot_params(seed=41, randomize_bn=True)
    dev = torch.device("cuda")
    sample = Voxelizer(**cfg)(pts)
    got = {}
    for field in (True, False):
        net = LisecNet(nx, ny, 8, 35, params=ParamStore(dev, init=op))
        net.field_conv = field
        cls, reg = net.forward(sample, training=training)
        torch.cuda.synchronize()
        assert net._used_field == field and (("grid" in net.act) == (not field))
        got[field] = dict(y=net.act["mid1.y"].cpu().numpy(), bn=net.bnstate["mid1.bn"].cpu().numpy(),
                          mm=net.params.view("mid1.bn.moving_mean").cpu().numpy(),
                          mv=net.params.view("mid1.bn.moving_variance").cpu().numpy(),
                          cls=cls.cpu().numpy(), reg=reg.cpu().numpy())
    f, d = got[True], got[False]
    scale = np.abs(d["y"]).max()
    assert np.abs(f["y"] - d["y"]).max() <= 2e-5 * scale, np.abs(f["y"] - d["y"]).max() / scale
    for k in ("bn", "mm", "mv"):
        np.testing.assert_allclose(f[k], d[k], rtol=2e-5, atol=2e-6 * max(1.0, np.abs(d[k]).max()), err_msg=k)
    close(f["cls"], d["cls"], rtol=1e-4, what="class map, field vs dense first Conv3D")
    close(f["reg"], d["reg"], rtol=1e-4, what="regression map, field vs dense first Conv3D")


def test_field_conv_is_deterministic_and_training_step_matches_dense_path():
    """Two runs of the field form give the same bits (fixed summation orders, fixed-point statistics), and one whole
    training step (loss, every gradient) agrees with the step that takes the dense first Conv3D."""
    from conftest import LYFT
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M

    op = M.glorot_params(seed=43, randomize_bn=True)
    dev = torch.device("cuda")
    sample = Voxelizer(**LYFT)(u20k(13))
    rng = np.random.default_rng(3)
    y_cls = torch.from_numpy(rng.integers(0, 2, (100, 200, 2)).astype(np.float32)).to(dev)
    y_reg = torch.from_numpy(rng.normal(0, 1, (100, 200, 14)).astype(np.float32)).to(dev)
    runs = []
    for field in (True, True, False):
        net = LisecNet(200, 400, 8, 35, params=ParamStore(dev, init=op))
        net.field_conv = field
        lo = net.train_step(sample, y_cls, y_reg)
        torch.cuda.synchronize()
        runs.append((net.act["mid1.y"].clone(), net.bnstate["mid1.bn"].clone(), lo.clone(), net.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert abs(runs[0][2][0].item() - runs[2][2][0].item()) <= 1e-5 * abs(runs[2][2][0].item())
    g_f, g_d = runs[0][3].double(), runs[2][3].double()
    # whole-gradient L2 distance between the two evaluation orders: fp32 noise amplified by the BN backward of nearly
    # constant maps, well under the 3e-3 the oracle comparison allows
    assert float((g_f - g_d).norm() / g_d.norm()) < 3e-3


@pytest.mark.parametrize("grid", ["small", "lyft"])
def test_collapsed_head_equals_layered_head(grid):
    """Conv2DTranspose x3 -> Concatenate -> two 1x1 heads (model_training.py:246-255) is linear from the branch inputs to
    the maps, so the default schedule contracts each branch with the composite kernel W_b . H_b straight to 16 channels
    (csrc/head_fused.hip).  It must give the maps, the loss and EVERY gradient of the layer-by-layer schedule
    (compose_head=False: 256-channel upsampling into the concat, 768 -> 16 heads) up to fp32 summation order."""
    from lisec_amd.network import LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    from oracle import model_ref as M

    dev = torch.device("cuda")
    op = M.glorot_params(seed=77, randomize_bn=True)
    for k in op:                                   # non-zero biases everywhere, so the composite bias is exercised
        if k.endswith(".bias"):
            op[k] = torch.from_numpy(np.random.default_rng(len(k)).normal(0, 0.05, tuple(op[k].shape)).astype(np.float32))
    if grid == "small":
        dims, cfg, pts = (16, 32, 8), SMALL, small_cloud(seed=5)
    else:
        from bench import u20k_cloud
        from lisec_amd import Constants
        dims = (Constants.nx, Constants.ny, Constants.nz)
        cfg = dict(xSize=Constants.voxelx, ySize=Constants.voxely, zSize=Constants.voxelz, sampleSize=35,
                   maxVoxelX=Constants.nx // 2, maxVoxelY=Constants.ny // 2, maxVoxelZ=Constants.nz)
        pts = u20k_cloud(3)
    Ho, Wo = dims[0] // 2, dims[1] // 2
    rng = np.random.default_rng(2)
    y_cls = torch.from_numpy(rng.integers(0, 3, (Ho, Wo, 2)).astype(np.float32)).to(dev)
    y_reg = torch.from_numpy(rng.normal(0, 1, (Ho, Wo, 14)).astype(np.float32)).to(dev)
    out = {}
    for compose in (False, True):
        net = LisecNet(dims[0], dims[1], dims[2], 35, params=ParamStore(dev, init=op), compose_head=compose)
        sample = Voxelizer(**cfg)(pts)
        for training in (False, True):
            cls, reg = net.forward(sample, training=training)
            out[(compose, training)] = (cls.cpu().numpy().copy(), reg.cpu().numpy().copy())
        lo = net.backward(y_cls, y_reg, loss="smoothl1_ce")
        torch.cuda.synchronize()
        out[(compose, "loss")] = lo.cpu().numpy().copy()
        out[(compose, "grad")] = {n: net.params.grad_view(net.grad, n).cpu().numpy().copy()
                                  for n in net.params.trainable_names()}
        del net
    for training in (False, True):
        for got, ref, what in zip(out[(True, training)], out[(False, training)], ("cls", "reg")):
            err = np.linalg.norm((got - ref).ravel()) / np.linalg.norm(ref.ravel())
            assert err < 2e-6, f"{what} (training={training}): relative L2 {err:.2e}"
    assert np.allclose(out[(True, "loss")], out[(False, "loss")], rtol=1e-6)
    worst = {}
    for n, ref in out[(False, "grad")].items():
        got = out[(True, "grad")][n]
        scale = np.linalg.norm(ref.ravel())
        if scale < 1e-12:
            assert np.abs(got).max() < 1e-6, n
            continue
        worst[n] = np.linalg.norm((got - ref).ravel()) / scale
    # the two schedules only differ downstream of the branch inputs; upstream layers see the same dL/dx_b up to rounding,
    # which the BatchNormalization backward of a 98 %-constant map amplifies: 2e-3 there, 1e-5 at the
    # branches and heads themselves
    for n, e in worst.items():
        tight = n.startswith(("up", "cls", "reg"))
        assert e < (2e-5 if tight else 5e-3), f"{n}: relative L2 {e:.2e}"
