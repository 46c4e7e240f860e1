"""CPU: the iou3d oracle (oracle/iou3d_oracle.c, a C restatement of iou3d_kernel.cu / iou3d.cpp) is pinned to
tests/golden/iou3d_ref.npz — outputs of the REFERENCE's own kernels (hipified into oracle/_ref, run on an MI355X by
tests/golden/make_iou3d_golden.py) — and cross-checked against an independent float64 polygon clipping."""
import os

import numpy as np

import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "iou3d_ref.npz")


def test_pairwise_matches_reference_kernel_outputs():
    z = np.load(GOLDEN)
    ov = oracle.iou3d_pairwise(z["boxes_a"], z["boxes_b"], "overlap")
    iou = oracle.iou3d_pairwise(z["boxes_a"], z["boxes_b"], "iou")
    assert int((z["overlap"] > 0).sum()) > 500                       # the fixture exercises real intersections
    assert np.max(np.abs(ov - z["overlap"])) <= 2e-5                 # libm vs device sin/cos/atan2 roundings
    assert np.max(np.abs(iou - z["iou"])) <= 2e-6
    assert np.array_equal(ov > 0, z["overlap"] > 0)


def test_nms_matches_reference_kernel_outputs():
    z = np.load(GOLDEN)
    for thr in ("0.1", "0.5"):
        assert np.array_equal(oracle.iou3d_nms(z["nms_boxes"], float(thr)), z[f"nms_keep_{thr}"])
        assert np.array_equal(oracle.iou3d_nms(z["nms_boxes"], float(thr), normal=True), z[f"nms_normal_keep_{thr}"])


def test_overlap_agrees_with_float64_clipping():
    rng = np.random.default_rng(1)
    c, wh = rng.uniform(-8, 8, (60, 2)), rng.uniform(0.5, 5, (60, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(-3.2, 3.2, (60, 1))], 1).astype(np.float32)
    ov = oracle.iou3d_pairwise(boxes[:30], boxes[30:], "overlap")
    ref = np.array([[oracle.rotated_overlap_float64(a, b) for b in boxes[30:]] for a in boxes[:30]])
    assert np.max(np.abs(ov - ref)) <= 5e-5 and (ref > 0).sum() > 50


def test_known_answers():
    sq = np.array([[0, 0, 2, 2, 0.0]], np.float32)
    assert abs(oracle.iou3d_pairwise(sq, sq, "iou")[0, 0] - 1.0) < 1e-6
    half = np.array([[1, 0, 3, 2, 0.0]], np.float32)                  # half overlap: 2 / (4 + 4 - 2)
    assert abs(oracle.iou3d_pairwise(sq, half, "iou")[0, 0] - 1 / 3) < 1e-6
    rot = np.array([[0, 0, 2, 2, np.pi / 4]], np.float32)             # octagon: 8 (sqrt 2 - 1)
    assert abs(oracle.iou3d_pairwise(sq, rot, "overlap")[0, 0] - 8 * (np.sqrt(2) - 1)) < 1e-5
    far = np.array([[10, 10, 11, 11, 0.3]], np.float32)
    assert oracle.iou3d_pairwise(sq, far, "overlap")[0, 0] == 0.0
    assert oracle.iou3d_pairwise(np.zeros((0, 5), np.float32), sq, "iou").shape == (0, 1)
