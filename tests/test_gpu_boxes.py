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
