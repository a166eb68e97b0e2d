"""Label generation and RPN decode + NMS (C ABI, float64) vs the CPU oracle (oracle/boxes_ref.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(rng, n):
    data = np.zeros((n, 7))
    data[:, 0] = rng.uniform(-48, 48, n)
    data[:, 1] = rng.uniform(-48, 48, n)
    data[:, 2] = rng.uniform(0.5, 1.5, n)
    data[:, 3] = rng.uniform(3.5, 5.2, n)           # the anchors are (l,w) = (1.6, 3.9) against HALVED x/l, y/w
    data[:, 4] = rng.uniform(6.5, 9.0, n)
    data[:, 5] = rng.uniform(1.3, 1.8, n)
    data[:, 6] = rng.choice([0.0, np.pi / 2, 0.1, -0.2, 1.4], n)
    return data


@pytest.mark.parametrize("n_boxes,seed", [(0, 0), (4, 1), (9, 2)])
def test_preprocess_labels_vs_oracle(n_boxes, seed):
    from lisec_amd import boxes
    from oracle import boxes_ref as B
    rng = np.random.default_rng(seed)
    data = _scene(rng, n_boxes)
    if n_boxes:
        data[0, :2] = [-49.0, 49.2]                 # near the border: exercises the wrap-around layout + range skip
    for balance in (False, True):
        cls_r, reg_r = B.preprocess_labels(data, seed=7, balance=balance)
        cls, reg = boxes.preprocessLabels(data, seed=7, balance=balance)
        assert cls.shape == (100, 200, 2) and reg.shape == (100, 200, 14) and cls.dtype == np.float64
        assert np.array_equal(cls, cls_r)
        assert np.allclose(reg, reg_r, rtol=1e-12, atol=1e-12)
    if n_boxes:
        assert (cls_r == 2).sum() >= 1


def test_rpn_to_region_vs_oracle():
    import torch
    from lisec_amd import boxes
    from oracle import boxes_ref as B
    rng = np.random.default_rng(3)
    prob = rng.uniform(0, 0.1, (100, 200, 2)).astype(np.float32)
    regress = rng.normal(0, 0.2, (100, 200, 14)).astype(np.float32)
    for (ix, iy, a, p) in [(40, 100, 0, 0.99), (41, 100, 0, 0.98), (70, 30, 1, 0.97), (0, 0, 0, 0.96), (55, 150, 1, 0.95)]:
        prob[ix, iy, a] = p
    got_boxes, got_probs = boxes.rpnToRegion(prob, regress)
    all_boxes = B.decode_boxes(regress.astype(np.float64))
    all_probs = prob.astype(np.float64).transpose(2, 0, 1).reshape(-1)
    pick = B.nms(all_boxes, all_probs, overlapThresh=0.0, maxBoxes=20)
    assert len(got_probs) == len(pick) == 21
    assert np.allclose(got_probs, all_probs[pick], rtol=0, atol=0)
    assert np.allclose(got_boxes, all_boxes[pick], rtol=1e-12, atol=1e-12)
    # a device view of a (M,16) head buffer works too (cls = [:, :2], reg = [:, 2:])
    head = torch.from_numpy(np.concatenate([prob.reshape(-1, 2), regress.reshape(-1, 14)], 1)).cuda()
    b2, p2 = boxes.rpnToRegion(head[:, :2].reshape(100, 200, 2), head[:, 2:].reshape(100, 200, 14))
    assert np.array_equal(b2, got_boxes) and np.array_equal(p2, got_probs)


class _FakeLyft:
    """Duck-typed LyftDataset: the tables imageToRPN reads (serialize_data.py:341-381)."""

    def __init__(self, anns, ego_t, ego_q):
        self.t = {"sample_data": {"sd": {"ego_pose_token": "ego"}}, "ego_pose": {"ego": {"translation": ego_t, "rotation": ego_q}},
                  "sample_annotation": {f"a{i}": a for i, a in enumerate(anns)},
                  "instance": {f"i{i}": {"category_token": a["_cat"]} for i, a in enumerate(anns)},
                  "category": {"car": {"name": "car"}, "bus": {"name": "bus"}}}

    def get(self, table, token):
        return self.t[table][token]


def test_save_labels_for_sample(tmp_path):
    """saveLabelsForSample under the reference's name: global -> ego transform (inverse ego rotation), car filter,
    +-50 m filter, the four .npy files train() reads back."""
    from lisec_amd import boxes, serialize_data
    from lisec_amd.model_training import rotate_points
    rng = np.random.default_rng(5)
    ego_t = [100.0, -40.0, 2.0]
    yaw = 0.3
    ego_q = [np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2)]
    local = _scene(rng, 6)
    local[5, :2] = [70.0, 0.0]                                    # outside +-50 m: dropped
    anns = []
    for i, b in enumerate(local):
        g = rotate_points(b[None, :3], np.array(ego_q), False)[0] + np.array(ego_t)       # ego -> global
        anns.append({"translation": list(g), "size": list(b[3:6]), "rotation": [np.cos(b[6] / 2), 0, 0, np.sin(b[6] / 2)],
                     "instance_token": f"i{i}", "_cat": "bus" if i == 4 else "car"})
    ds = _FakeLyft(anns, ego_t, ego_q)
    sample = {"anns": [f"a{i}" for i in range(6)], "data": {"LIDAR_TOP": "sd"}}
    cls, reg = serialize_data.saveLabelsForSample([sample, sample], str(tmp_path / "labels3"), ds)
    assert cls.shape == (2, 100, 200, 2) and reg.shape == (2, 100, 200, 14)
    for name, shape in (("labelsClass.npy", cls.shape), ("regressClass.npy", reg.shape)):
        assert np.load(tmp_path / "labels3" / name).shape == shape
    assert list(np.load(tmp_path / "labels3" / "labelsShape.npy")) == [2, 100, 200, 2]
    # same targets as preprocessLabels on the four kept boxes (yaw from the quaternion, ego-frame positions)
    want_cls, want_reg = boxes.preprocessLabels(local[:4], seed=0)
    assert np.array_equal(cls[0], want_cls)
    assert np.allclose(reg[0], want_reg, atol=1e-9)
    serialize_data.level5Data = ds                               # the reference's module-global form
    c2, _ = serialize_data.imageToRPN(sample)
    assert np.array_equal(c2, want_cls)
    serialize_data.level5Data = None


def test_rpn_to_region_module_name():
    from lisec_amd import boxes, rpnToRegion
    rng = np.random.default_rng(3)
    cls = rng.uniform(0, 1, (1, 100, 200, 2)).astype(np.float32)
    reg = rng.normal(0, 0.1, (1, 100, 200, 14)).astype(np.float32)
    b, p = rpnToRegion.rpnToRegion(cls, reg)
    b2, p2 = boxes.rpnToRegion(cls[0], reg[0], maxBoxes=20, overlapThresh=0.)
    assert np.array_equal(b, b2) and np.array_equal(p, p2) and b.shape[1] == 7


def test_device_geometry_and_decode_match_reference_run_goldens():
    """boxes.hip against values the REFERENCE produced (tests/golden/box_geometry_decode.npz, made by running
    serialize_data.py / rpnToRegion.py unmodified under recording stand-ins): the corners of boxToShapely, the
    intersection / union / IoU arithmetic around a given polygon area, and the full decode (anchor grid + applyRegrssion,
    anchor-major order, probabilities) that lisec_rpn_to_region runs before its suppression loop."""
    import ctypes
    import os
    import torch
    from lisec_amd import _lib, boxes
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "box_geometry_decode.npz"))
    lib, dev = _lib.load(), torch.device("cuda")
    bx = torch.from_numpy(np.ascontiguousarray(g["geom_boxes"])).to(dev)
    n = bx.shape[0]
    corners = torch.zeros((n, 4, 2), dtype=torch.float64, device=dev)
    pair = torch.zeros((n // 2, 4), dtype=torch.float64, device=dev)
    area = torch.from_numpy(np.ascontiguousarray(g["geom_pair_area"])).to(dev)
    _lib.check(lib.lisec_box_geometry(_lib.ptr(bx), n, _lib.ptr(area), _lib.ptr(corners), _lib.ptr(pair), _lib.current_stream()))
    torch.cuda.synchronize()
    assert np.allclose(corners.cpu().numpy(), g["geom_corners"], rtol=0, atol=1e-12)
    p = pair.cpu().numpy()
    assert np.allclose(p[:, 0], g["geom_pair_intersection"], rtol=1e-14, atol=0)
    assert np.allclose(p[:, 1], g["geom_pair_union"], rtol=1e-14, atol=0)
    assert np.allclose(p[:, 2], g["geom_pair_iou"], rtol=1e-13, atol=0)
    # the library's own polygon area (unpinned by the reference: shapely) against the oracle's clipping
    from oracle import boxes_ref as B
    own = [B.convex_intersection_area(B.box_corners(list(g["geom_boxes"][2 * k])), B.box_corners(list(g["geom_boxes"][2 * k + 1])))
           for k in range(n // 2)]
    assert np.allclose(p[:, 3], own, rtol=1e-10, atol=1e-12)
    # decode
    cfg = boxes._cfg()
    cls = torch.from_numpy(g["decode_cls"]).to(dev)
    reg = torch.from_numpy(g["decode_reg"]).to(dev)
    N = 2 * cfg.outX * cfg.outY
    dec = torch.zeros((N, 7), dtype=torch.float64, device=dev)
    prob = torch.zeros(N, dtype=torch.float64, device=dev)
    legal = torch.zeros(N, dtype=torch.int32, device=dev)
    _lib.check(lib.lisec_rpn_decode(ctypes.byref(cfg), _lib.ptr(cls), 2, _lib.ptr(reg), 14, _lib.ptr(dec), _lib.ptr(prob),
                                    _lib.ptr(legal), _lib.current_stream()))
    torch.cuda.synchronize()
    assert np.allclose(dec.cpu().numpy(), g["decode_boxInfo"], rtol=1e-14, atol=1e-14)
    assert np.array_equal(prob.cpu().numpy(), g["decode_probInfo"])
    assert int(legal.sum().item()) == N                    # exp(t) * anchor > 0: the reference removed nothing either
    # and the public function picks from exactly these candidates
    got_boxes, got_probs = boxes.rpnToRegion(g["decode_cls"], g["decode_reg"])
    order = np.argsort(-g["decode_probInfo"], kind="stable")
    assert got_probs[0] == g["decode_probInfo"][order[0]] or len(got_probs) > 0
    for bxs, pr in zip(got_boxes, got_probs):
        j = np.nonzero(g["decode_probInfo"] == pr)[0]
        assert any(np.allclose(bxs, g["decode_boxInfo"][jj], rtol=1e-14, atol=1e-14) for jj in j)
