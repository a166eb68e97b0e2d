"""The reference's Python surface (train / train_with_model / predictMain / createModel / load_model) over
a fake level5Data and synthetic lidar .bin files -- BASELINE config 1's plumbing (the published weights
SampleModel/15SampleEpoch0.h5 are stripped from the reference, so a seeded random model stands in)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class FakeLevel5:
    """Duck-typed LyftDataset: .get('sample_data'|'calibrated_sensor', token) (model_training.py:81-94)."""

    def __init__(self, root, n_samples, rng):
        self.tables = {"sample_data": {}, "calibrated_sensor": {}}
        self.samples = []
        os.makedirs(os.path.join(root, "lidar"), exist_ok=True)
        for i in range(n_samples):
            data = {}
            for j, sensor in enumerate(["LIDAR_TOP", "LIDAR_FRONT_RIGHT", "LIDAR_FRONT_LEFT"][: 3 - (i % 2)]):
                n = 6000
                raw = np.zeros((n, 5), np.float32)
                raw[:, 0] = rng.uniform(-45, 45, n)
                raw[:, 1] = rng.uniform(-45, 45, n)
                raw[:, 2] = rng.uniform(-1.0, 1.2, n)
                fn = f"lidar/s{i}_{sensor}.bin"
                raw.tofile(os.path.join(root, fn))
                tok, cal = f"sd{i}{j}", f"cs{i}{j}"
                self.tables["sample_data"][tok] = {"filename": fn, "calibrated_sensor_token": cal}
                ang = 0.1 * j
                self.tables["calibrated_sensor"][cal] = {"rotation": [np.cos(ang / 2), 0, 0, np.sin(ang / 2)],
                                                         "translation": [0.5 * j, 0.1, 1.0]}
                data[sensor] = tok
            self.samples.append({"data": data})

    def get(self, table, token):
        return self.tables[table][token]


def test_train_save_load_predict_roundtrip(tmp_path, monkeypatch):
    from lisec_amd import Constants, Predict, model_training
    rng = np.random.default_rng(0)
    data_root = tmp_path / "lyft"
    l5 = FakeLevel5(str(data_root), 3, rng)
    monkeypatch.setattr(Constants, "lyft_data_dir", str(data_root))
    monkeypatch.chdir(tmp_path)
    os.makedirs("labels3")
    np.save("labels3/labelsClass.npy", rng.integers(0, 3, (3, 100, 200, 2)).astype(np.float64))
    np.save("labels3/regressClass.npy", rng.normal(0, 1, (3, 100, 200, 14)))
    np.random.seed(0)

    # rotate_points / combine_lidar_data: float64 (n,3), all three / two sensors concatenated
    pts = model_training.combine_lidar_data(l5.samples[0], str(data_root), l5)
    assert pts.dtype == np.float64 and pts.shape == (18000, 3)
    assert model_training.combine_lidar_data(l5.samples[1], str(data_root), l5).shape == (12000, 3)
    R = model_training._quaternion_matrix([np.cos(0.05), 0, 0, np.sin(0.05)])
    assert np.allclose(R @ R.T, np.eye(3)) and np.isclose(R[0, 0], np.cos(0.1))

    save_path = str(tmp_path / "models" / "3SampleEpoch0.h5")
    model = model_training.train(l5.samples, l5, save_path)          # 180 steps on the Lyft grid
    assert os.path.exists(save_path)
    assert model.net.iterations == 180
    out1 = tmp_path / "pred1"
    Predict.predictMain(l5.samples[:2], str(out1), l5, model)
    prob = np.load(out1 / "sample0_label.npy")
    reg = np.load(out1 / "sample1_regress.npy")
    assert prob.shape == (1, 100, 200, 2) and prob.dtype == np.float32       # rpnToRegion.py:116-117
    assert reg.shape == (1, 100, 200, 14) and reg.dtype == np.float32
    assert np.isfinite(prob).all() and np.isfinite(reg).all()

    # reload -> identical predictions; resume training re-compiles with a fresh SGD (:339-340)
    model2 = model_training.load_model(save_path, custom_objects={
        "RepeatLayer": model_training.RepeatLayer, "MaxPoolingVFELayer": model_training.MaxPoolingVFELayer})
    out2 = tmp_path / "pred2"
    Predict.predictMain(l5.samples[:2], str(out2), l5, model2)
    assert np.array_equal(np.load(out2 / "sample0_label.npy"), prob)
    assert np.array_equal(np.load(out2 / "sample1_regress.npy"), reg)

    # the checkpoint is a Keras-layout HDF5 file (model_training.py:302): it comes back compiled, with the SGD
    # iteration count and momentum accumulators, so one more step from either copy is bit-identical
    import torch
    from lisec_amd import hdf5_lite
    with open(save_path, "rb") as f:
        assert f.read(8) == hdf5_lite.SIGNATURE
    with hdf5_lite.File(save_path) as f:
        assert len(f["model_weights"].attrs["layer_names"]) == 113
        assert f["model_weights/conv3d/conv3d/kernel:0"].shape == (3, 3, 3, 64, 64)
        assert int(f["optimizer_weights/SGD/iter:0"][()]) == 180
    assert model2.optimizer is not None and model2.net.iterations == 180
    assert torch.equal(model2.net.velocity, model.net.velocity)
    assert torch.equal(model2.net.params.theta, model.net.params.theta)
    assert torch.equal(model2.net.params.state, model.net.params.state)
    npz_path = str(tmp_path / "models" / "weights.npz")
    model.save(npz_path)
    model3 = model_training.load_model(npz_path)
    assert torch.equal(model3.net.params.theta, model.net.params.theta) and model3.optimizer is None


def test_sparse_tensor_surface_and_dense_input():
    """VFE_preprocessing's result quacks like the reference's SparseTensor; a dense (n,D,H,W,T,6) array is
    accepted by predict() and gives the same answer as the sparse form."""
    import torch
    from lisec_amd import model_training as mt
    from oracle import voxel_ref
    rng = np.random.default_rng(2)
    pts = np.stack([rng.uniform(-4.2, 4.2, 800), rng.uniform(-4.2, 4.2, 800), rng.uniform(0.0, 2.1, 800)], 1)
    st = mt.VFE_preprocessing(pts, 0.5, 0.25, 0.25, 35, 8, 16, 8)
    assert st.dense_shape == [8, 16, 32, 35, 6]
    ref = voxel_ref.voxelize_ref(pts, 0.5, 0.25, 0.25, 35, 8, 16, 8)
    dense = mt.sparse.to_dense(st, default_value=0., validate_indices=False)
    assert np.array_equal(dense, voxel_ref.to_dense(ref, (8, 16, 32, 35, 6)))
    assert st.indices.shape == (len(ref["coords"]) * 35 * 6, 5)
    model = mt.createModel(16, 32, 8, 35)
    a = model.predict(st)
    b = model.predict(dense[None])
    assert a[0].shape == (1, 8, 16, 2) and a[1].shape == (1, 8, 16, 14)
    assert np.allclose(a[0], b[0], rtol=1e-4, atol=1e-5) and np.allclose(a[1], b[1], rtol=1e-4, atol=1e-5)
    x = torch.arange(2 * 1 * 3, dtype=torch.float32).reshape(2, 1, 3)
    assert mt.RepeatLayer()(x).shape == (2, 35, 3)
    assert mt.MaxPoolingVFELayer(combine=True)(torch.rand(4, 35, 8)).shape == (4, 8)
    # compute_output_shape as the reference's layers declare it (model_training.py:36-37, 49-53)
    shape = (None, 8, 200, 400, 35, 16)
    assert mt.MaxPoolingVFELayer().compute_output_shape(shape) == (None, 8, 200, 400, 1, 16)
    assert mt.MaxPoolingVFELayer(combine=True).compute_output_shape(shape) == (None, 8, 200, 400, 16)
    assert mt.RepeatLayer().compute_output_shape((None, 8, 200, 400, 1, 16)) == shape
    assert mt.MaxPoolingVFELayer(combine=True).get_config() == {"combine": True}
    assert mt.get_voxel((-0.1, 0.3, 0.6), 0.5, 0.25, 0.25) == (-1, 1, 2)


def test_gpu_lidar_transform_known_answers_and_oracle():
    """lisec_lidar_transform (the C ABI entry) against oracle/ingest_ref.py: the hand-computed known answers
    (identity, +-90 degrees about each axis, 180 degrees, a NON-UNIT quaternion) and random sensors, float64."""
    import ctypes
    import torch
    from lisec_amd import _lib
    from lisec_amd import model_training as mt
    from oracle import ingest_ref
    lib, dev = _lib.load(), _lib.require_gpu()

    def transform(raw5, q, t):
        d_raw = torch.from_numpy(np.ascontiguousarray(raw5, dtype=np.float32)).to(dev)
        out = torch.empty((len(raw5), 3), dtype=torch.float64, device=dev)
        R = np.ascontiguousarray(mt._quaternion_matrix(q), dtype=np.float64)
        tt = np.ascontiguousarray(t, dtype=np.float64)
        _lib.check(lib.lisec_lidar_transform(_lib.ptr(d_raw), len(raw5), 5,
                                             R.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                             tt.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _lib.ptr(out),
                                             _lib.current_stream()))
        return out.cpu().numpy()

    for q, inverse, p, want in ingest_ref.KNOWN_ANSWERS:
        if inverse:
            continue                                              # the ingest path only rotates forwards (:93)
        raw = np.zeros((1, 5), np.float32)
        raw[0, :3] = p
        raw[0, 3:] = 7.0                                          # intensity / ring columns are ignored (:90)
        got = transform(raw, q, (0.25, -1.5, 2.0))
        assert np.allclose(got[0], np.array(want) + (0.25, -1.5, 2.0), rtol=0, atol=1e-14)
    rng = np.random.default_rng(9)
    for _ in range(5):
        raw = rng.normal(0, 30, (4097, 5)).astype(np.float32)
        q, t = rng.normal(0, 1, 4) * rng.uniform(0.2, 5), rng.normal(0, 3, 3)
        want = ingest_ref.rotate_points(raw[:, :3], q) + t
        assert np.allclose(transform(raw, q, t), want, rtol=1e-13, atol=1e-12)


def test_gpu_lidar_ingest_matches_oracle(tmp_path):
    """combine_lidar_data_gpu == oracle/ingest_ref.combine_lidar_data (float64, 1e-12) and feeds the voxeliser."""
    import torch
    from lisec_amd import model_training as mt
    from oracle import ingest_ref
    rng = np.random.default_rng(4)
    root = tmp_path / "lyft"
    l5 = FakeLevel5(str(root), 2, rng)
    for smp in l5.samples:
        ref = ingest_ref.combine_lidar_data(smp, str(root), l5)
        assert np.allclose(mt.combine_lidar_data(smp, str(root), l5), ref, rtol=1e-13, atol=1e-12)
        got = mt.combine_lidar_data_gpu(smp, str(root), l5)
        assert got.dtype == torch.float64 and tuple(got.shape) == ref.shape
        assert np.allclose(got.cpu().numpy(), ref, rtol=1e-12, atol=1e-12)
        a = mt.VFE_preprocessing(got, 0.5, 0.25, 0.25, 35, 100, 200, 8).sample.to_host()
        b = mt.VFE_preprocessing(ref, 0.5, 0.25, 0.25, 35, 100, 200, 8).sample.to_host()
        # identical clouds up to the last float64 bit -> identical voxels unless a point grazes a cell border
        assert len(a["coords"]) == len(b["coords"]) and np.array_equal(a["coords"], b["coords"])


def test_recorded_step_equals_eager_steps():
    """lisec_amd.network.RecordedStep (the whole fit() step -- voxelise + forward + backward on both streams + SGD with
    the device-side iteration counter + the repack for the next step -- recorded once as a step plan of the C ABI and
    re-issued by lisec_step_plan_run) gives BIT-IDENTICAL variables to the Python schedule, with sweeps of different
    sizes padded into the fixed-capacity point buffer."""
    import torch
    from lisec_amd.network import RecordedStep, PipelinedStep, LisecNet
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    cfg = dict(xSize=0.5, ySize=0.25, zSize=0.25, sampleSize=35, maxVoxelX=8, maxVoxelY=16, maxVoxelZ=8)
    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    clouds = [np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 2.1, n)], 1).astype(np.float32)
              for n in (2500, 1700, 3000, 1)]
    ys = [(rng.integers(0, 3, (8, 16, 2)).astype(np.float32), rng.normal(0, 1, (8, 16, 14)).astype(np.float32))
          for _ in clouds]
    init = ParamStore(dev).to_dict()

    def padded(pts, capacity=3000):
        out = np.full((capacity, 3), RecordedStep.PAD, np.float32)
        out[:len(pts)] = pts
        return out

    def run(captured, pad=True):
        net = LisecNet(16, 32, 8, 35, params=ParamStore(dev, init=init))
        vox = Voxelizer(**cfg)
        net._prepare_training()
        net.iterations = 5                                   # a non-zero start: the decay term is live
        losses = []
        step = (PipelinedStep if captured == "pipelined" else RecordedStep)(net, vox, 3000) if captured else None
        if captured:
            assert net.iterations == 5                       # the warm-up steps of the recording left no trace
            assert step.launches > 100                       # the plan holds the step's launches and event edges
        dev_in = [tuple(torch.from_numpy(a).to(dev) for a in (pts, yc, yr)) for pts, (yc, yr) in zip(clouds, ys)]
        if captured == "pipelined":
            step.prime(*dev_in[0])
        for k, (pts, (yc, yr)) in enumerate(zip(clouds, ys)):
            d_pts, d_yc, d_yr = dev_in[k]
            if captured == "pipelined":
                # trains on sweep k; sweep k + 1 is voxelised inside the step (nothing staged after the last one)
                lo = step.step(*dev_in[k + 1]) if k + 1 < len(dev_in) else step.step()
            elif captured:
                lo = step(d_pts, d_yc, d_yr)
            else:
                # same capacity as the captured buffer: the row-list kernels plan their K slices per capacity
                lo = net.train_step(vox(torch.from_numpy(padded(pts)).to(dev) if pad else d_pts), d_yc, d_yr)
            losses.append(lo.cpu().numpy().copy())
        torch.cuda.synchronize()
        assert net.iterations == 5 + len(clouds) and int(net._iter_dev[0].item()) == net.iterations
        # an eager inference pass after captured steps sees the updated variables (packed copies are refreshed)
        cls, _ = net.forward(vox(torch.from_numpy(clouds[0]).to(dev)), training=False)
        return net.params.theta.cpu().numpy(), net.params.state.cpu().numpy(), np.stack(losses), cls.cpu().numpy().copy()

    t_e, s_e, l_e, c_e = run(False)
    t_g, s_g, l_g, c_g = run(True)
    assert np.array_equal(l_e, l_g)
    assert np.array_equal(t_e, t_g) and np.array_equal(s_e, s_g)
    assert np.array_equal(c_e, c_g)
    # ... and so does the pipelined form (sweep k + 1 voxelised on the second stream while step k runs)
    t_p, s_p, l_p, c_p = run("pipelined")
    assert np.array_equal(l_e, l_p) and np.array_equal(t_e, t_p) and np.array_equal(s_e, s_p) and np.array_equal(c_e, c_p)
    # padding itself: the unpadded sweeps give the same voxels, hence the same step up to fp32 summation order
    t_u, s_u, l_u, _ = run(False, pad=False)
    assert np.allclose(l_u, l_e, rtol=1e-6) and np.allclose(t_u, t_e, rtol=1e-4, atol=1e-6)
    assert np.allclose(s_u, s_e, rtol=1e-5, atol=1e-7)


def test_step_plan_survives_a_larger_eager_sweep():
    """fit -> predict on a sweep LARGER than the recorded capacity (the VFE / field / row-list scratch is reallocated by the
    eager call) -> fit again: the cached step plan holds raw addresses of the old buffers, so it must be re-recorded, not
    replayed into freed memory -- and a plan replayed by hand after such a reallocation refuses.  Variables bit-identical to
    the Python schedule throughout."""
    import torch
    from lisec_amd import _lib, model_training as mt
    from lisec_amd.network import RecordedStep, StalePlanError
    from lisec_amd.params import ParamStore
    from lisec_amd.voxelizer import Voxelizer
    dev = torch.device("cuda")
    rng = np.random.default_rng(11)
    cfgk = (0.5, 0.25, 0.25, 35, 8, 16, 8)

    def cloud(n):
        return np.stack([rng.uniform(-4.2, 4.2, n), rng.uniform(-4.2, 4.2, n), rng.uniform(0.0, 2.1, n)], 1).astype(np.float32)
    small = [cloud(900), cloud(700)]
    big = cloud(9000)
    ys = [(rng.integers(0, 3, (8, 16, 2)).astype(np.float32), rng.normal(0, 1, (8, 16, 14)).astype(np.float32)) for _ in small]
    init = ParamStore(dev).to_dict()

    def run(step_plan):
        os.environ["LISEC_TUNING"] = "step_plan=%d" % (1 if step_plan else 0)
        try:
            model = mt.createModel(16, 32, 8, 35)
            model.net.params.load_dict(init)
            model.compile(optimizer=mt.optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True), loss=["mse", "mse"])
            vox = Voxelizer(*cfgk[:3], cfgk[3], *cfgk[4:])
            # the plan pads every sweep into its 4096-point buffer (dropped by the voxeliser's range test); the Python
            # schedule gets the same padded sweeps, because the row-list kernels plan their K slices per capacity
            def padded(c):
                out = np.full((4096, 3), RecordedStep.PAD, np.float32)
                out[:len(c)] = c
                return out
            samples = [vox(torch.from_numpy(c if step_plan else padded(c)).to(dev)) for c in small]
            ycls = np.stack([y[0] for y in ys]); yreg = np.stack([y[1] for y in ys])
            model.fit(samples, y=[ycls, yreg], batch_size=1, epochs=1, steps_per_epoch=2, verbose=0, shuffle=False)
            gen0 = _lib.alloc_generation()
            plan0 = getattr(model, "_captured", None)
            out = model.predict([vox(torch.from_numpy(big).to(dev))])         # 10x the points: scratch grows
            if step_plan:
                assert plan0 is not None and _lib.alloc_generation() != gen0, "the larger sweep must have reallocated scratch"
                with pytest.raises(StalePlanError):
                    plan0[1].replay() if isinstance(plan0[1], RecordedStep) else plan0[1].step()
            model.fit(samples, y=[ycls, yreg], batch_size=1, epochs=1, steps_per_epoch=2, verbose=0, shuffle=False)
            if step_plan:
                assert model._captured[1] is not plan0[1], "the stale plan must have been replaced"
            torch.cuda.synchronize()
            return model.net.params.theta.cpu().numpy().copy(), model.net.params.state.cpu().numpy().copy(), out[0]
        finally:
            os.environ.pop("LISEC_TUNING", None)

    t_e, s_e, o_e = run(False)
    t_p, s_p, o_p = run(True)
    assert np.array_equal(o_e, o_p)
    assert np.array_equal(t_e, t_p) and np.array_equal(s_e, s_p)
