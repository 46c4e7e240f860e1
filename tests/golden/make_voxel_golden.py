"""Generates tests/golden/voxel_ref_*.npz with the REFERENCE's own hard_voxelize_cpu
(oracle/_ref/voxel_layer, compiled from /root/reference/mmdet3d/ops/voxel/src by oracle/ref_build.py).

Runs in the build container (CPU).  Only CUBIC grids: the reference's CPU code indexes its lookup
grid [x][y][z] but allocates it [gz,gy,gx] (voxelization_cpu.cpp:129-130 vs :75,83) and is
memory-unsafe otherwise.

    python tests/golden/make_voxel_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_build  # noqa: E402

CASES = {
    # name: (n_points, grid, max_points, max_voxels, features, seed)
    "a": (20000, 40, 10, 160000, 5, 0),   # neither cap binds much
    "b": (20000, 40, 3, 5000, 5, 1),      # both caps bind
    "c": (3000, 16, 2, 100, 4, 2),        # tiny, heavy truncation, F=4
}


def main():
    ext = ref_build.load_ref("voxel_layer")
    for name, (n, g, mp, mv, f, seed) in CASES.items():
        rng = np.random.default_rng(seed)
        lo, hi = -4.0, 4.0
        pts = rng.uniform(lo - 0.5, hi + 0.5, size=(n, f)).astype(np.float32)  # some points fall outside
        pts[::7, :3] = pts[3, :3]  # duplicates: one crowded voxel
        vs = [(hi - lo) / g] * 3
        cr = [lo, lo, lo, hi, hi, hi]
        p = torch.from_numpy(pts)
        voxels = torch.zeros(mv, mp, f)
        coors = torch.zeros(mv, 3, dtype=torch.int32)
        npv = torch.zeros(mv, dtype=torch.int32)
        m = ext.hard_voxelize(p, voxels, coors, npv, vs, cr, mp, mv, 3, True)
        dyn = torch.zeros(n, 3, dtype=torch.int32)
        ext.dynamic_voxelize(p, dyn, vs, cr, 3)
        out = os.path.join(HERE, f"voxel_ref_{name}.npz")
        np.savez_compressed(out, points=pts, voxel_size=np.array(vs, np.float32), coors_range=np.array(cr, np.float32),
                            max_points=mp, max_voxels=mv, voxels=voxels[:m].numpy(), coors=coors[:m].numpy(),
                            num_points_per_voxel=npv[:m].numpy(), dynamic_coors=dyn.numpy())
        print(name, "voxel_num", m, "->", out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
