"""Generates tests/golden/scatter_ref.npz with the REFERENCE's own dynamic-scatter kernels.

`dynamic_point_to_voxel_forward / _backward` have no CPU path in the reference (voxelization.h:118,139), so this runs on
the GPU box: oracle/_ref/voxel_layer/voxel_layer.so is the reference's extension, hipified from /root/reference at build
time by oracle/ref_build.py (sources never enter this repository).

    gpurun -- python tests/golden/make_scatter_golden.py gpurun_out/golden
then copy gpurun_out/golden/scatter_ref.npz into tests/golden/.

Feature values are multiples of 1/8 so that the reference's atomicAdd sums are exact in any order; the fixture therefore
pins sum / mean bit for bit, not only max.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_build  # noqa: E402

# name: (num_points, ndim, coordinate upper bounds, channels, share of rows with a negative entry, seed)
CASES = {
    "small3": (400, 3, (6, 5, 4), 5, 0.1, 0),
    "dense3": (3000, 3, (4, 4, 3), 3, 0.0, 1),      # ~60 points per voxel, many exact ties for max
    "wide3": (5000, 3, (1440, 1440, 41), 4, 0.05, 2),
    "batch4": (2000, 4, (3, 20, 18, 5), 6, 0.02, 3),
    "allbad": (50, 3, (4, 4, 4), 2, 1.0, 4),
}


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    ext = ref_build.load_ref("voxel_layer")
    dev = torch.device("cuda:0")
    out = {}
    for name, (n, ndim, hi, c, bad, seed) in CASES.items():
        rng = np.random.default_rng(seed)
        coors = np.stack([rng.integers(0, h, n) for h in hi], 1).astype(np.int32)
        neg = rng.random(n) < bad
        coors[neg, rng.integers(0, ndim, int(neg.sum()))] = -1
        feats = (np.round(rng.standard_normal((n, c)) * 8) / 8).astype(np.float32)
        out[f"{name}.coors"], out[f"{name}.feats"] = coors, feats
        for mode in ("sum", "mean", "max"):
            red, oc, cmap, cnt = ext.dynamic_point_to_voxel_forward(torch.from_numpy(feats).to(dev),
                                                                    torch.from_numpy(coors).to(dev), mode)
            g = (np.round(rng.standard_normal(tuple(red.shape)) * 8) / 8).astype(np.float32)
            gf = torch.zeros(n, c, device=dev)
            ext.dynamic_point_to_voxel_backward(gf, torch.from_numpy(g).to(dev), torch.from_numpy(feats).to(dev), red, cmap,
                                                cnt, mode)
            torch.cuda.synchronize()
            for k, v in (("reduced", red), ("out_coors", oc), ("coors_map", cmap), ("count", cnt), ("grad_feats", gf)):
                out[f"{name}.{mode}.{k}"] = v.cpu().numpy()
            out[f"{name}.{mode}.grad_reduced"] = g
        print(name, "N", n, "M", int(out[f"{name}.max.out_coors"].shape[0]))
    path = os.path.join(out_dir, "scatter_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/golden")
