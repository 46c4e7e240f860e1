"""Generates tests/golden/voxel_gpu_ref.npz with the REFERENCE's own GPU hard voxelization (`hard_voxelize` on device
tensors, deterministic=True -> hard_voxelize_gpu, mmdet3d/ops/voxel/src/voxelization_cuda.cu:231-373), run on an MI355X from
oracle/_ref/voxel_layer/voxel_layer.so (the reference's extension hipified from /root/reference at build time by
oracle/ref_build.py — sources never enter this repository).  VERDICT r4 item 4: the GPU voxelizer was pinned by the CPU functor and
by reading; this pins it to what the reference's CUDA op itself returns.

    gpurun -- python tests/golden/make_voxel_gpu_golden.py gpurun_out/golden
then copy gpurun_out/golden/voxel_gpu_ref.npz into tests/golden/.

Cases: the inputs of the three CPU goldens (voxel_ref_{a,b,c}.npz, cubic grids) and one sweep of the synthetic LiDAR on the real
1440 x 1440 x 40 grid (voxel 0.075 / 0.075 / 0.2 m), uncapped and with both caps binding.  The O(N^2) duplicate scan of the
reference (point_to_voxelidx_kernel) bounds the cloud size; one sweep = 31 k points.  The big `voxels` tensors of the LiDAR cases
are stored as SHA-256 + the per-voxel feature SUMS in float64 (order-independent check) to keep the fixture small."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from bevfusion_amd import synth  # noqa: E402
from oracle import ref_build  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(ext, pts, vs, cr, mp, mv, dev):
    p = torch.from_numpy(pts).to(dev)
    voxels = torch.zeros(mv, mp, pts.shape[1], device=dev)
    coors = torch.zeros(mv, 3, dtype=torch.int32, device=dev)
    npv = torch.zeros(mv, dtype=torch.int32, device=dev)
    m = ext.hard_voxelize(p, voxels, coors, npv, [float(v) for v in vs], [float(v) for v in cr], mp, mv, 3, True)
    torch.cuda.synchronize()
    dyn = torch.zeros(pts.shape[0], 3, dtype=torch.int32, device=dev)
    ext.dynamic_voxelize(p, dyn, [float(v) for v in vs], [float(v) for v in cr], 3)
    torch.cuda.synchronize()
    return int(m), voxels[:m].cpu().numpy(), coors[:m].cpu().numpy(), npv[:m].cpu().numpy(), dyn.cpu().numpy()


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    ext = ref_build.load_ref("voxel_layer")
    dev = torch.device("cuda:0")
    out = {}
    for name in "abc":
        z = np.load(os.path.join(GOLDEN, f"voxel_ref_{name}.npz"))
        m, v, c, n, dyn = run(ext, z["points"], z["voxel_size"], z["coors_range"], int(z["max_points"]), int(z["max_voxels"]), dev)
        out[f"{name}.voxels"], out[f"{name}.coors"], out[f"{name}.num_points_per_voxel"], out[f"{name}.dynamic_coors"] = v, c, n, dyn
        print(name, "voxel_num", m, "cpu golden", z["coors"].shape[0],
              "coors equal cpu golden:", bool(np.array_equal(c, z["coors"])), "reversed:", bool(np.array_equal(c[:, ::-1], z["coors"])),
              "voxels equal:", bool(v.shape == z["voxels"].shape and np.array_equal(v, z["voxels"])))
    cfg = synth.CL_CONFIG
    pts = synth.lidar_points(seed=0, sweeps=1)
    out["lidar.points_sha256"] = np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), np.uint8)
    out["lidar.num_points"] = np.int64(pts.shape[0])
    for tag, mp, mv in (("lidar", cfg["max_num_points"], cfg["max_voxels"][1]), ("lidar_capped", 2, 5000)):
        m, v, c, n, dyn = run(ext, pts, cfg["voxel_size"], cfg["point_cloud_range"], mp, mv, dev)
        out[f"{tag}.max_points"], out[f"{tag}.max_voxels"] = np.int64(mp), np.int64(mv)
        out[f"{tag}.coors"], out[f"{tag}.num_points_per_voxel"] = c, n
        out[f"{tag}.voxels_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(v).tobytes()).digest(), np.uint8)
        out[f"{tag}.voxel_sums"] = v.astype(np.float64).sum(1).astype(np.float32)
        if tag == "lidar":
            out["lidar.dynamic_coors_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(dyn).tobytes()).digest(), np.uint8)
        print(tag, "N", pts.shape[0], "voxel_num", m, "max count", int(n.max()))
    path = os.path.join(out_dir, "voxel_gpu_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/golden")
