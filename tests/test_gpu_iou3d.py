"""GPU parity of the HIP iou3d (csrc/iou3d.hip through the C ABI and the reference-named Python API) against the golden
outputs of the reference's own kernels and against the CPU oracle.

Bars: overlap / IoU within 1e-4 (fp32 geometry with device sin/cos/atan2; north_star tolerance for float outputs);
NMS keep sets bit-exact on the fixtures (no pair sits within rounding of a threshold); device sweep == oracle sweep."""
import os

import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import iou3d

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "iou3d_ref.npz")


def _boxes(rng, n, spread):
    c, wh = rng.uniform(-spread, spread, (n, 2)), rng.uniform(0.4, 6.0, (n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)


def test_pairwise_vs_reference_golden(dev):
    z = np.load(GOLDEN)
    a, b = torch.from_numpy(z["boxes_a"]).to(dev), torch.from_numpy(z["boxes_b"]).to(dev)
    ov = iou3d.boxes_overlap_bev(a, b).cpu().numpy()
    iou = iou3d.boxes_iou_bev(a, b).cpu().numpy()
    assert np.max(np.abs(ov - z["overlap"])) <= 1e-4 and np.max(np.abs(iou - z["iou"])) <= 1e-4
    # drop-in module: writes into caller tensors
    out = torch.zeros((a.shape[0], b.shape[0]), device=dev)
    assert iou3d.iou3d_cuda.boxes_iou_bev_gpu(a, b, out) == 1
    assert torch.equal(out.cpu(), torch.from_numpy(iou))


@pytest.mark.parametrize("m,n", [(1, 1), (3, 200), (65, 64), (130, 257)])
def test_pairwise_vs_oracle_random(dev, m, n):
    rng = np.random.default_rng(m * 1000 + n)
    a, b = _boxes(rng, m, 8.0), _boxes(rng, n, 8.0)
    got = iou3d.boxes_iou_bev(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    assert np.max(np.abs(got - oracle.iou3d_pairwise(a, b, "iou"))) <= 1e-4
    got = iou3d.boxes_overlap_bev(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy()
    assert np.max(np.abs(got - oracle.iou3d_pairwise(a, b, "overlap"))) <= 1e-4


def test_nms_vs_reference_golden(dev):
    z = np.load(GOLDEN)
    boxes = torch.from_numpy(z["nms_boxes"]).to(dev)
    scores = torch.arange(boxes.shape[0], 0, -1, device=dev, dtype=torch.float32)   # already in score order
    for thr in ("0.1", "0.5"):
        assert np.array_equal(iou3d.nms_gpu(boxes, scores, float(thr)).cpu().numpy(), z[f"nms_keep_{thr}"])
        assert np.array_equal(iou3d.nms_normal_gpu(boxes, scores, float(thr)).cpu().numpy(), z[f"nms_normal_keep_{thr}"])
        keep = torch.zeros(boxes.shape[0], dtype=torch.long)
        k = iou3d.iou3d_cuda.nms_gpu(boxes, keep, float(thr), 0)
        assert np.array_equal(keep[:k].numpy(), z[f"nms_keep_{thr}"])


@pytest.mark.parametrize("n,spread,thr", [(1, 5.0, 0.3), (63, 4.0, 0.3), (64, 4.0, 0.2), (65, 4.0, 0.2), (1000, 9.0, 0.25),
                                          (5000, 40.0, 0.1)])
def test_nms_vs_oracle_random_with_scores_and_limits(dev, n, spread, thr):
    rng = np.random.default_rng(n)
    b = _boxes(rng, n, spread)
    s = rng.random(n).astype(np.float32)
    order = np.argsort(-s, kind="stable")
    want = order[oracle.iou3d_nms(b[order], thr)]
    got = iou3d.nms_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), thr).cpu().numpy()
    assert np.array_equal(got, want)
    want_n = order[oracle.iou3d_nms(b[order], thr, normal=True)]
    assert np.array_equal(iou3d.nms_normal_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), thr).cpu().numpy(), want_n)
    if n >= 64:
        pre = n // 2
        want_p = order[:pre][oracle.iou3d_nms(b[order[:pre]], thr)][:10]
        got_p = iou3d.nms_gpu(torch.from_numpy(b).to(dev), torch.from_numpy(s).to(dev), thr, pre_maxsize=pre, post_max_size=10)
        assert np.array_equal(got_p.cpu().numpy(), want_p)
    # device-resident count variant
    keep, cnt = iou3d.nms_sorted(torch.from_numpy(b[order]).to(dev), thr, sync=False)
    assert int(cnt) == want.shape[0] and np.array_equal(order[keep[: int(cnt)].cpu().numpy()], want)


def test_empty_and_bad_inputs(dev):
    e = torch.zeros((0, 5), device=dev)
    one = torch.tensor([[0, 0, 1, 1, 0.0]], device=dev)
    assert tuple(iou3d.boxes_iou_bev(e, one).shape) == (0, 1)
    assert iou3d.nms_sorted(e, 0.5).numel() == 0
    with pytest.raises(RuntimeError, match="GPU tensor"):
        iou3d.boxes_iou_bev(torch.zeros(2, 5), one)
    with pytest.raises(RuntimeError, match=r"\[N, 5\]"):
        iou3d.boxes_iou_bev(torch.zeros((2, 7), device=dev), one)
