"""Drop-in counterpart of the reference's model_training.py (same public names and argument order):

    get_voxel, VFE_preprocessing, combine_lidar_data, rotate_points, RepeatLayer, MaxPoolingVFELayer,
    createModel, load_model, optimizers.SGD, train, train_with_model

The Keras graph is replaced by lisec_amd.network.LisecNet (HIP kernels behind the C ABI); lidar sweeps
stay sparse on the GPU instead of being densified to (8,200,400,35,6) and stacked in host RAM
(reference model_training.py:279,285).  There is no CPU fallback.
"""
import json
import os
import time

import numpy as np
import torch

from . import Constants, _lib
from .network import LisecNet
from .params import ParamStore
from .voxelizer import VoxelSample, Voxelizer, host_row_stats


# ---------------------------------------------------------------------------------------------------
# voxeliser front end (reference model_training.py:103-152)
def get_voxel(point, xSize, ySize, zSize):
    """Voxel coordinate of a point; voxels are named by their lower corner (model_training.py:103-107)."""
    from math import floor
    return (floor(point[0] / xSize), floor(point[1] / ySize), floor(point[2] / zSize))


_VOXELIZERS = {}


def VFE_preprocessing(points, xSize, ySize, zSize, sampleSize, maxVoxelX, maxVoxelY, maxVoxelZ):
    """points (n, >=3) -> SparseVoxels, the stand-in for the tf.SparseTensor of dense_shape
    [maxVoxelZ, 2*maxVoxelX, 2*maxVoxelY, sampleSize, 6] the reference returns (model_training.py:112-152).
    Deterministic: a voxel holding more than sampleSize points keeps the lowest point indices."""
    key = (float(xSize), float(ySize), float(zSize), int(sampleSize), int(maxVoxelX), int(maxVoxelY), int(maxVoxelZ))
    if key not in _VOXELIZERS:
        _VOXELIZERS[key] = Voxelizer(*key[:3], key[3], *key[4:])
    return SparseVoxels(_VOXELIZERS[key](points))


class SparseVoxels:
    """Quacks like the SparseTensor the reference builds: .indices (z,x,y,t,f), .values, .dense_shape
    (materialised on the host only when asked for); carries the device-resident VoxelSample the network eats."""

    def __init__(self, sample):
        self.sample = sample
        c = sample.cfg
        self.dense_shape = [c.maxVoxelZ, 2 * c.maxVoxelX, 2 * c.maxVoxelY, c.sampleSize, 6]
        self.shape = tuple(self.dense_shape)
        self._coo = None

    def _materialise(self):
        if self._coo is None:
            h = self.sample.to_host()
            V, T = len(h["coords"]), self.sample.cfg.sampleSize
            idx = np.empty((V, T, 6, 5), dtype=np.int64)
            idx[..., :3] = h["coords"][:, None, None, :]
            idx[..., 3] = np.arange(T)[None, :, None]
            idx[..., 4] = np.arange(6)[None, None, :]
            self._coo = (idx.reshape(-1, 5), h["feats"].reshape(-1))
        return self._coo

    @property
    def indices(self):
        return self._materialise()[0]

    @property
    def values(self):
        return self._materialise()[1]


class sparse:   # noqa: N801  (mirrors `from tensorflow import sparse`)
    @staticmethod
    def to_dense(st, default_value=0., validate_indices=False):
        """tf.sparse.to_dense (model_training.py:279): host numpy array of st.dense_shape.  Only meant for
        small grids / inspection -- the network consumes the sparse form directly."""
        dense = np.full(st.dense_shape, default_value, dtype=np.float32)
        idx, val = st.indices, st.values
        dense[tuple(idx.T)] = val
        return dense

    @staticmethod
    def reshape(st, shape):
        return st            # (1,) + shape: the batch axis is implicit (Predict.py:29)


def dense_to_sample(dense, device=None):
    """(D,H,W,T,6) dense array -> VoxelSample.  Every row of a non-empty voxel is kept as a real row
    (zero rows are numerically identical to pad rows), so this is exact for ANY dense input."""
    device = device or _lib.require_gpu()
    dense = np.asarray(dense, dtype=np.float32)
    D, H, W, T, F = dense.shape
    if F != 6:
        raise ValueError("last axis must be 6")
    flat = dense.reshape(D * H * W, T, 6)
    occ = np.nonzero(np.abs(flat).reshape(len(flat), -1).max(1) > 0)[0]
    V = len(occ)
    cfg = _lib.VoxelCfg(1.0, 1.0, 1.0, H // 2, W // 2, D, T)
    cell_voxel = np.full(D * H * W, -1, np.int32)
    cell_voxel[occ] = np.arange(V, dtype=np.int32)
    coords = np.stack([occ // (H * W), (occ // W) % H, occ % W], 1).astype(np.int32)
    npts = np.full(V, T, np.int32)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)  # noqa: E731
    n_rows = max(V * T, 1)
    rows = np.zeros((n_rows, 6), np.float32)
    rows[:V * T] = flat[occ].reshape(-1, 6)
    s = VoxelSample(cfg, n_rows, max(V, 1), t(np.array([V, V * T, V * T, T, 0, 0, 0, 0]), torch.int32),
                    t(cell_voxel, torch.int32), t(coords if V else np.zeros((1, 3)), torch.int32),
                    t(npts if V else np.zeros(1), torch.int32), t(npts if V else np.zeros(1), torch.int32),
                    t(np.arange(max(V, 1) + 1) * T, torch.int32), t(rows, torch.float32),
                    t(np.arange(n_rows), torch.int32), t(host_row_stats(rows[:V * T]), torch.int64))
    return s


# ---------------------------------------------------------------------------------------------------
# lidar assembly (reference model_training.py:65-98)
def _quaternion_matrix(q):
    w, x, y, z = (float(v) for v in q)
    n = (w * w + x * x + y * y + z * z) ** 0.5
    w, x, y, z = w / n, x / n, y / n, z / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rotate_points(points, rotation, inverse=False):
    """Rotate by a (w,x,y,z) quaternion (model_training.py:65-69)."""
    R = _quaternion_matrix(rotation)
    if inverse:
        R = R.T
    return np.dot(R, np.asarray(points).T).T


def combine_lidar_data(sample, dataDir, level5Data):
    """Every lidar point of a sample in the car frame, float64 (n,3) (model_training.py:73-98)."""
    sensorTypes = ['LIDAR_TOP', 'LIDAR_FRONT_RIGHT', 'LIDAR_FRONT_LEFT']
    actual = [s for s in sensorTypes if s in sample['data']]     # not every sample has all three (:75-79)
    allPoints = []
    for sensorType in actual:
        frame = level5Data.get('sample_data', sample['data'][sensorType])
        sensor = level5Data.get('calibrated_sensor', frame['calibrated_sensor_token'])
        filePath = os.path.join(dataDir, *frame['filename'].replace('\\', '/').split('/'))
        raw = np.fromfile(filePath, dtype=np.float32).reshape(-1, 5)[:, :3]          # :87-90
        pts = rotate_points(raw, sensor['rotation']) + np.array(sensor['translation'])   # :93-94
        allPoints.append(pts)
    return np.concatenate(allPoints)


def combine_lidar_data_gpu(sample, dataDir, level5Data, device=None):
    """combine_lidar_data with the rotate + translate + concatenate done on the GPU: the raw float32 .bin rows
    are uploaded once (20 B/point) and the float64 (n,3) cloud never exists on the host.  Returns a device
    tensor that VFE_preprocessing accepts directly."""
    import ctypes
    device = device or _lib.require_gpu()
    lib = _lib.load()
    sensorTypes = ['LIDAR_TOP', 'LIDAR_FRONT_RIGHT', 'LIDAR_FRONT_LEFT']
    frames = []
    for sensorType in [s for s in sensorTypes if s in sample['data']]:
        frame = level5Data.get('sample_data', sample['data'][sensorType])
        sensor = level5Data.get('calibrated_sensor', frame['calibrated_sensor_token'])
        filePath = os.path.join(dataDir, *frame['filename'].replace('\\', '/').split('/'))
        raw = np.fromfile(filePath, dtype=np.float32).reshape(-1, 5)
        frames.append((raw, sensor))
    total = sum(len(r) for r, _ in frames)
    out = torch.empty((total, 3), dtype=torch.float64, device=device)
    at = 0
    for raw, sensor in frames:
        d_raw = torch.from_numpy(raw).to(device)
        R = np.ascontiguousarray(_quaternion_matrix(sensor['rotation']), dtype=np.float64)
        t = np.ascontiguousarray(np.asarray(sensor['translation'], dtype=np.float64))
        _lib.check(lib.lisec_lidar_transform(
            _lib.ptr(d_raw), len(raw), 5, R.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            t.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_void_p(out.data_ptr() + at * 24),
            _lib.current_stream()))
        at += len(raw)
    return out


# ---------------------------------------------------------------------------------------------------
# the two custom layers the reference names in custom_objects (model_training.py:32-61)
class RepeatLayer:
    """repeat_elements(x, maxPoints, axis=-2): (…,1,C) -> (…,35,C)."""

    def __init__(self, **kwargs):
        pass

    def compute_output_shape(self, inputShape):
        """(…, 1, C) -> (…, maxPoints, C) (reference model_training.py:36-37)."""
        inputShape = tuple(inputShape)
        return inputShape[:Constants.pointIndex] + (Constants.maxPoints,) + inputShape[Constants.pointIndex + 1:]

    def __call__(self, inputs):
        return torch.repeat_interleave(torch.as_tensor(inputs), Constants.maxPoints, dim=Constants.pointIndex)

    call = __call__


class MaxPoolingVFELayer:
    """max over the point axis, keepdims unless combine=True; pad rows take part (no mask)."""

    def __init__(self, combine=False, **kwargs):
        self.combineDim = combine

    def compute_output_shape(self, inputShape):
        """(…, T, C) -> (…, 1, C), or (…, C) with combine=True (reference model_training.py:49-53)."""
        inputShape = tuple(inputShape)
        mid = () if self.combineDim else (1,)
        return inputShape[:Constants.pointIndex] + mid + inputShape[Constants.pointIndex + 1:]

    def __call__(self, inputs):
        return torch.as_tensor(inputs).max(dim=Constants.pointIndex, keepdim=not self.combineDim).values

    call = __call__

    def get_config(self):
        return {'combine': self.combineDim}


# ---------------------------------------------------------------------------------------------------
class optimizers:   # noqa: N801  (mirrors `from tensorflow.keras import optimizers`)
    class SGD:
        def __init__(self, lr=0.01, decay=0.0, momentum=0.0, nesterov=False, learning_rate=None):
            self.lr = learning_rate if learning_rate is not None else lr
            self.decay, self.momentum, self.nesterov = decay, momentum, nesterov
            if not nesterov:
                raise NotImplementedError("only the reference's configuration (nesterov=True) is implemented")


class History:
    def __init__(self):
        self.history = {}


class Model:
    """What createModel returns: the subset of the keras.Model interface the reference uses."""

    def __init__(self, nx, ny, nz, maxPoints, params=None):
        self.nx, self.ny, self.nz, self.maxPoints = nx, ny, nz, maxPoints
        self.net = LisecNet(nx, ny, nz, maxPoints, params=params)
        self.optimizer, self.loss = None, None
        self.dp = None
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            from .parallel import DataParallel
            self.dp = DataParallel(self.net.device)
            self.dp.broadcast_(self.net.params.theta)
            self.dp.broadcast_(self.net.params.state)
            self.net.params.touch()

    # -- compile / fit / predict / save ------------------------------------------------------------
    def compile(self, optimizer, loss):
        """compile(optimizer=sgd, loss=['mse','mse']) (model_training.py:296).  A fresh optimizer discards
        velocity and iteration count, as re-compiling does in the reference (:339-340)."""
        if isinstance(loss, (list, tuple)):
            kinds = [str(x).lower() for x in loss]
            if kinds == ['mse', 'mse']:
                loss = 'mse'
            elif kinds == ['cross_entropy', 'smooth_l1'] or kinds == ['smoothl1_ce']:
                loss = 'smoothl1_ce'
            else:
                raise ValueError(f"unsupported loss list {loss}")
        if loss not in ('mse', 'smoothl1_ce'):
            raise ValueError(f"unsupported loss {loss}")
        self.optimizer, self.loss = optimizer, loss
        self.net._prepare_training()
        self.net.velocity.zero_()
        self.net.iterations = 0

    def _as_samples(self, x):
        if isinstance(x, SparseVoxels):
            return [x.sample]
        if isinstance(x, VoxelSample):
            return [x]
        if isinstance(x, (list, tuple)):
            return [s for item in x for s in self._as_samples(item)]
        arr = np.asarray(x)
        if arr.ndim == 5:
            arr = arr[None]
        if arr.ndim != 6:
            raise ValueError("expected SparseVoxels / list of them / dense (n,D,H,W,T,6) array")
        return [dense_to_sample(a, self.net.device) for a in arr]

    def fit(self, x, y, batch_size=1, verbose=1, epochs=1, steps_per_epoch=None, shuffle=True):
        """fit(x=trainPoints, y=[outClass, outRegress], batch_size=1, epochs=1, steps_per_epoch=180)
        (model_training.py:299).  batch_size must be 1 (the reference's setting: BatchNormalization
        statistics are per sample).  With WORLD_SIZE > 1 whole samples are sharded over the ranks and the
        gradients averaged with one RCCL all-reduce per step."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model first")
        if batch_size != 1:
            raise NotImplementedError("only batch_size=1, the reference's setting, is implemented")
        samples = self._as_samples(x)
        ycls = np.asarray(y[0], dtype=np.float32)
        yreg = np.asarray(y[1], dtype=np.float32)
        n = len(samples)
        if len(ycls) < n or len(yreg) < n:
            raise ValueError("fewer label maps than samples")
        idx = list(range(n))
        if self.dp is not None:
            idx = self.dp.shard(idx)
        steps = steps_per_epoch if steps_per_epoch is not None else len(idx)
        if self.dp is not None and steps_per_epoch is not None:
            steps = max(1, steps_per_epoch // self.dp.world)
        dev = self.net.device
        hist = History()
        o = self.optimizer
        # targets live on the device for the whole fit when they fit comfortably (1.28 MB per sample)
        on_dev = n * ycls[0].size * 4 * 8 < (2 << 30)
        if on_dev:
            ycls_d = torch.from_numpy(np.ascontiguousarray(ycls[:n])).to(dev)
            yreg_d = torch.from_numpy(np.ascontiguousarray(yreg[:n])).to(dev)
        target = (lambda i: (ycls_d[i], yreg_d[i])) if on_dev else (
            lambda i: (torch.from_numpy(np.ascontiguousarray(ycls[i])).to(dev),
                       torch.from_numpy(np.ascontiguousarray(yreg[i])).to(dev)))
        captured = self._captured_step(samples)
        for _ in range(epochs):
            order = list(np.random.permutation(idx)) if shuffle else list(idx)
            # the running loss stays on the device: reading it back every step would stall the host behind the GPU and
            # expose the time it needs to enqueue the next step; the progress line is refreshed ~20 times per epoch
            tot_dev = torch.zeros(3, dtype=torch.float64, device=dev)
            every = max(1, steps // 20)
            t0 = time.time()
            for st in range(steps):
                i = int(order[st % len(order)])
                yc, yr = target(i)
                if captured is not None and hasattr(captured, "prime"):
                    # ... pipelined: this step also voxelises the sweep of the next one (second stream, under the backward)
                    if st == 0:
                        captured.prime(samples[i]._keepalive, yc, yr)
                    if st + 1 < steps:
                        j = int(order[(st + 1) % len(order)])
                        captured.step(samples[j]._keepalive, *target(j))
                    else:
                        captured.step()               # (the next epoch draws its own order and primes its first sweep)
                elif captured is not None:
                    # the whole step (voxelise + forward + backward + update) re-issued from its recorded plan: one C call
                    captured(samples[i]._keepalive, yc, yr)
                else:
                    self.net.forward(samples[i], training=True)
                    if self.dp is not None:
                        avg = self.dp.bucketed()
                        self.net.backward(yc, yr, loss=self.loss,
                                          rpn_grads_ready=lambda lo, hi, avg=avg: avg.start_tail(self.net.grad, lo, hi))
                        avg.finish(self.net.grad)
                    else:
                        self.net.backward(yc, yr, loss=self.loss, rpn_grads_ready=lambda lo, hi: self.net.early_update(
                            lo, hi, lr=o.lr, decay=o.decay, momentum=o.momentum))
                    self.net.apply_gradients(lr=o.lr, decay=o.decay, momentum=o.momentum)
                tot_dev += self.net.loss_out
                if verbose and ((st + 1) % every == 0 or st + 1 == steps):
                    print(f"\r{st + 1}/{steps} - loss: {float(tot_dev[0].item()) / (st + 1):.4f}", end="", flush=True)
            tot = tot_dev.cpu().numpy()
            if verbose:
                print(f" - {time.time() - t0:.1f}s")
            for key, v in zip(("loss", "ClassificationLayer_loss", "RegressionLayer_loss"), tot / max(steps, 1)):
                hist.history.setdefault(key, []).append(float(v))
        return hist

    def _captured_step(self, samples):
        """The recorded form of the step (lisec_amd.network.RecordedStep: the eager schedule re-issued by
        lisec_step_plan_run, one C call per step), when it applies: one GPU, every sample a voxelised sweep that still
        holds its device points, one grid (data parallel included: the gradient exchange is recorded with the step).
        Otherwise (None) the Python schedule issues every step."""
        if not _lib.knob("step_plan", True) or not samples:
            return None
        pts = [getattr(s, "_keepalive", None) for s in samples]
        if any(p is None or not p.is_cuda for p in pts):
            return None                       # e.g. dense arrays turned into samples: no point cloud to re-voxelise
        c0 = samples[0].cfg
        key0 = (c0.xSize, c0.ySize, c0.zSize, c0.sampleSize, c0.maxVoxelX, c0.maxVoxelY, c0.maxVoxelZ)
        for s_ in samples:
            c = s_.cfg
            if (c.xSize, c.ySize, c.zSize, c.sampleSize, c.maxVoxelX, c.maxVoxelY, c.maxVoxelZ) != key0:
                return None
        dtype = torch.float64 if any(p.dtype == torch.float64 for p in pts) else torch.float32
        need = max(int(p.shape[0]) for p in pts)
        o = self.optimizer
        key = (key0, dtype, self.loss, o.lr, o.decay, o.momentum, id(self.net), torch.cuda.current_stream().cuda_stream)
        cur = getattr(self, "_captured", None)
        if cur is not None and cur[0] == key and cur[1].capacity >= need and cur[1].alloc_gen == _lib.alloc_generation():
            return cur[1]
        if cur is not None:
            # another grid / optimizer / network -- or an eager call since (predict() on a larger sweep, a second model)
            # reallocated a workspace the plan holds the raw address of: the old plan points at dead buffers
            cur[1].close()
        from .network import RecordedStep, PipelinedStep
        capacity = max(1024, -(-need // 4096) * 4096)        # a little head-room: later fits reuse the plan
        step = (PipelinedStep if _lib.knob("pipeline_voxels", True) else RecordedStep)(self.net, Voxelizer(*key0[:3], key0[3], *key0[4:], device=self.net.device), capacity,
                            dtype=dtype, loss=self.loss, lr=o.lr, decay=o.decay, momentum=o.momentum,
                            allreduce=self.dp.bucketed() if self.dp is not None else None)
        self._captured = (key, step)
        return step

    def predict(self, x):
        """predict(testVFEPointsDense) -> [prob (n,Ho,Wo,2), regress (n,Ho,Wo,14)] (Predict.py:38);
        BatchNormalization uses the moving statistics."""
        probs, regs = [], []
        for s in self._as_samples(x):
            cls, reg = self.net.forward(s, training=False)
            probs.append(cls.cpu().numpy().copy())
            regs.append(reg.cpu().numpy().copy())
        return [np.concatenate(probs), np.concatenate(regs)]

    def save(self, path):
        """model.save(save_path) (model_training.py:302): a Keras-layout HDF5 file (model_config, model_weights/<layer>/
        <layer>/<weight>:0 with Keras' automatic layer names, training_config and the SGD iteration count + momentum
        accumulators under optimizer_weights), written by lisec_amd.hdf5_lite -- see lisec_amd/keras_h5.py.  A path
        ending in .npz gets a plain numpy archive with the names of lisec_amd.params.param_specs() instead."""
        p = self.net.params
        if self.dp is not None:
            # data parallel: theta is identical on every rank, the BatchNormalization moving statistics are per
            # replica (each rank saw its own samples).  The checkpoint carries their MEAN over the ranks and is
            # written by rank 0 alone; the replicas' own statistics are left as they are.
            own = p.state.clone()
            self.dp.average_(p.state)
            d = p.to_dict() if self.dp.rank == 0 else None
            p.state.copy_(own)
            if self.dp.rank != 0:
                self.dp.barrier()
                return
        else:
            d = p.to_dict()
        try:
            self._write(path, d)
        finally:
            if self.dp is not None:
                self.dp.barrier()          # nobody reads the file before rank 0 has closed it

    def _write(self, path, d):
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        if str(path).endswith(".npz"):
            meta = dict(format="lisec_amd-npz-1", nx=self.nx, ny=self.ny, nz=self.nz, maxPoints=self.maxPoints,
                        iterations=self.net.iterations)
            with open(path, "wb") as f:
                np.savez(f, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **d)
            return
        from . import keras_h5
        opt = vel = None
        if self.optimizer is not None:
            o = self.optimizer
            opt = dict(lr=o.lr, decay=o.decay, momentum=o.momentum, nesterov=o.nesterov)
            p = self.net.params
            vel = {n: p.view(n, buf=self.net.velocity).detach().cpu().numpy() for n in p.trainable_names()}
        keras_h5.save_model(path, d, self.nx, self.ny, self.nz, self.maxPoints, optimizer=opt,
                            iterations=self.net.iterations, velocity=vel)

    def summary(self):
        n = self.net.params.n_trainable()
        print(f"LisecNet grid ({self.nz},{self.nx},{self.ny},{self.maxPoints},6) -> "
              f"({self.nx // 2},{self.ny // 2},2) / ({self.nx // 2},{self.ny // 2},14); trainable params: {n:,}")


def createModel(nx, ny, nz, maxPoints):
    """createModel(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints) (model_training.py:222-257)."""
    return Model(nx, ny, nz, maxPoints)


def load_model(path, custom_objects=None):
    """load_model(model_path, custom_objects={'RepeatLayer':…, 'MaxPoolingVFELayer':…}) (:337-338, Predict.py:51-52).
    Reads Keras HDF5 files (the reference's own checkpoints or Model.save's) and the .npz variant.  Like Keras, a file
    that carries a training_config comes back compiled, with the saved iteration count and momentum accumulators."""
    with open(path, "rb") as f:
        magic = f.read(8)
    dev = _lib.require_gpu()
    if magic[:4] != b"\x89HDF":
        z = np.load(path, allow_pickle=False)
        meta = json.loads(bytes(z["__meta__"]).decode())
        params = ParamStore(dev, init={k: z[k] for k in z.files if k != "__meta__"})
        m = Model(meta["nx"], meta["ny"], meta["nz"], meta["maxPoints"], params=params)
        m.net.iterations = int(meta.get("iterations", 0))
        return m
    from . import keras_h5
    ck = keras_h5.load_model(path)
    m = Model(ck["nx"], ck["ny"], ck["nz"], ck["maxPoints"], params=ParamStore(dev, init=ck["params"]))
    o = ck["optimizer"]
    if o is not None and o.get("nesterov"):
        m.compile(optimizer=optimizers.SGD(lr=o["lr"], decay=o["decay"], momentum=o["momentum"], nesterov=True),
                  loss=['mse', 'mse'])
        m.net.iterations = ck["iterations"]
        if ck["velocity"] is not None:
            p = m.net.params
            for n in p.trainable_names():
                if n in ck["velocity"]:
                    p.view(n, buf=m.net.velocity).copy_(torch.from_numpy(np.ascontiguousarray(ck["velocity"][n])))
    return m


# ---------------------------------------------------------------------------------------------------
def _preprocess(samples, level5Data, dataDir):
    points = []
    for i in range(len(samples)):
        sampleLidarPoints = combine_lidar_data(samples[i], dataDir, level5Data)
        startTime = time.time()
        vfe_points = VFE_preprocessing(sampleLidarPoints, Constants.voxelx, Constants.voxely, Constants.voxelz,
                                       Constants.maxPoints, Constants.nx // 2, Constants.ny // 2, Constants.nz)
        points.append(vfe_points)       # stays sparse on the GPU: no to_dense, no 500 GB tf.stack (:279-285)
        print(time.time() - startTime)
        print('finished ' + str(i))
    return points


def _load_labels(labels_dir='labels3'):
    print('loading labels')
    outClass = np.load(os.path.join(labels_dir, 'labelsClass.npy'), allow_pickle=False)      # :289
    outRegress = np.load(os.path.join(labels_dir, 'regressClass.npy'), allow_pickle=False)   # :290
    return outClass, outRegress


def train(samples, level5Data, save_path):
    """train(samples, level5Data, save_path) (model_training.py:260-302)."""
    trainPoints = _preprocess(samples, level5Data, Constants.lyft_data_dir)
    outClass, outRegress = _load_labels()
    model = createModel(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints)
    sgd = optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True)
    model.compile(optimizer=sgd, loss=['mse', 'mse'])
    history = model.fit(x=trainPoints, y=[outClass, outRegress], batch_size=1, verbose=1, epochs=1,
                        steps_per_epoch=180)
    if model.dp is None or model.dp.rank == 0:
        print(history.history)
    model.save(save_path)
    return model


def train_with_model(samples, level5Data, model_path, save_path):
    """train_with_model(samples, level5Data, model_path, save_path) (model_training.py:305-346)."""
    trainPoints = _preprocess(samples, level5Data, Constants.lyft_data_dir)
    outClass, outRegress = _load_labels()
    model = load_model(model_path, custom_objects={'RepeatLayer': RepeatLayer,
                                                   'MaxPoolingVFELayer': MaxPoolingVFELayer})
    sgd = optimizers.SGD(lr=0.01, decay=1e-6, momentum=0.9, nesterov=True)
    model.compile(optimizer=sgd, loss=['mse', 'mse'])
    history = model.fit(x=trainPoints, y=[outClass, outRegress], batch_size=1, verbose=1, epochs=1,
                        steps_per_epoch=180)
    if model.dp is None or model.dp.rank == 0:
        print(history.history)
    model.save(save_path)
    return model


def _lyft_dataset():
    """The driver blocks of the reference build a LyftDataset from a hard-coded Windows path (model_training.py:349-355,
    Predict.py:43-49); here the root comes from Constants.lyft_data_dir ($LISEC_LYFT_DATA_DIR)."""
    try:
        from lyft_dataset_sdk.lyftdataset import LyftDataset
    except ImportError as e:                       # the SDK is not a dependency of the hot path
        raise SystemExit("the command-line driver needs lyft_dataset_sdk (pip install lyft-dataset-sdk): " + str(e))
    return LyftDataset(data_path=Constants.lyft_data_dir, json_path=os.path.join(Constants.lyft_data_dir, 'train_data'),
                       verbose=True)


if __name__ == '__main__':
    # python -m lisec_amd.model_training [save_path]      (model_training.py:349-364)
    import sys
    level5Data = _lyft_dataset()
    save_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join('models', '180SampleEpoch0.h5')
    samples = [level5Data.get('sample', scene['first_sample_token']) for scene in level5Data.scene]
    print('Training on ' + str(len(samples)))
    train(samples[:], level5Data, save_path)
