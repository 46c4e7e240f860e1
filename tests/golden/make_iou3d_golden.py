"""Generates tests/golden/iou3d_ref.npz with the REFERENCE's own iou3d kernels.

The reference has no CPU iou3d, so this runs on the GPU box: oracle/_ref/iou3d_cuda/iou3d_cuda.so is the reference's
extension, hipified from /root/reference at build time by oracle/ref_build.py (sources never enter this repository).

    gpurun -- python tests/golden/make_iou3d_golden.py gpurun_out/golden
then copy gpurun_out/golden/iou3d_ref.npz into tests/golden/.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_build  # noqa: E402


def random_boxes(rng, n, spread=12.0):
    c = rng.uniform(-spread, spread, (n, 2))
    wh = rng.uniform(0.4, 6.0, (n, 2))
    ang = rng.uniform(-np.pi, np.pi, (n, 1))
    return np.concatenate([c - wh / 2, c + wh / 2, ang], 1).astype(np.float32)


def special_boxes():
    b = [[0, 0, 4, 2, 0.0], [0, 0, 4, 2, 0.0],            # identical
         [1, 0.5, 3, 1.5, 0.3],                            # contained, rotated
         [4, 0, 8, 2, 0.0],                                # shares an edge with box 0
         [0, 0, 4, 2, np.pi / 2], [0, 0, 4, 2, np.pi / 4],  # same centre, rotated
         [10, 10, 11, 11, 1.0],                            # far away
         [-1, -1, 5, 3, -0.2], [2, -3, 3, 5, 0.0], [0, 0, 4, 2, 1e-3]]
    return np.asarray(b, np.float32)


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    ext = ref_build.load_ref("iou3d_cuda")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(20240807)
    a = np.concatenate([special_boxes(), random_boxes(rng, 86)])
    b = np.concatenate([special_boxes()[::-1], random_boxes(rng, 120)])
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    overlap = torch.zeros((a.shape[0], b.shape[0]), device=dev)
    iou = torch.zeros_like(overlap)
    ext.boxes_overlap_bev_gpu(ta, tb, overlap)
    ext.boxes_iou_bev_gpu(ta, tb, iou)
    # NMS inputs: dense clusters so that suppression happens, already sorted by (descending) score
    nb = random_boxes(rng, 300, spread=6.0)
    tn = torch.from_numpy(nb).to(dev)
    out = dict(boxes_a=a, boxes_b=b, overlap=overlap.cpu().numpy(), iou=iou.cpu().numpy(), nms_boxes=nb)
    for thr in (0.1, 0.5):
        keep = torch.zeros(nb.shape[0], dtype=torch.long)
        k = ext.nms_gpu(tn, keep, thr, 0)
        out[f"nms_keep_{thr}"] = keep[:k].numpy()
        keep = torch.zeros(nb.shape[0], dtype=torch.long)
        k = ext.nms_normal_gpu(tn, keep, thr, 0)
        out[f"nms_normal_keep_{thr}"] = keep[:k].numpy()
    torch.cuda.synchronize()
    np.savez_compressed(os.path.join(out_dir, "iou3d_ref.npz"), **out)
    print("wrote", os.path.join(out_dir, "iou3d_ref.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/golden")
