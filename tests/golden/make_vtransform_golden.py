"""Generates tests/golden/vtransform_ref.npz by running the REFERENCE's own Python function bodies
(`/root/reference/mmdet3d/models/vtransforms/base.py`: `gen_dx_bx` :15-21, `create_frustum` :66-89, `get_geometry`
:92-135, the cell-index / range-mask prologue of `BaseTransform.bev_pool` :149-169 and the depth raster of
`BaseDepthTransform.forward` :283-329) on CPU torch in this container.

`base.py` imports `mmcv.runner.force_fp32` and `mmdet3d.ops.bev_pool`, neither importable here: both are replaced by
inert stubs in `sys.modules` (force_fp32 -> identity decorator; bev_pool is never called because `bev_pool` / `get_cam_feats`
are overridden by a capturing subclass) and the file is exec'd unmodified from where it lies.  No reference source is copied.

torch runs single-threaded so that `index_put` on colliding pixels is sequential (= last point in input order wins;
with several threads the winner of a collision is unordered on CPU, as it is on GPU).

The 3x3 inverses inside these functions are `torch.inverse` = LAPACK (MKL sgetrf/sgetrs, un-vendored third-party
arithmetic): the fixture records the inverse matrices the reference computed, so that the op-by-op fp32 chain that follows
them can be pinned bit for bit.

Two cases, both with rotated / flipped image augmentation and a LiDAR augmentation (so every 3x3 product has a full matrix):
  small     2 samples x 6 cameras, dbound [1, 60, 2.0], feature 8x22: full geometry + cell indices + kept mask
  flagship  1 sample  x 6 cameras at the C+L sizes (N' = 1 993 728): SHA-256 of geometry / cell index / mask bytes
depth raster (both): 256x704 image, ~60k LiDAR points per sample: hit pixels (linear index, value), sparse.

    python tests/golden/make_vtransform_golden.py
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from bevfusion_amd import synth  # noqa: E402

REF_BASE = "/root/reference/mmdet3d/models/vtransforms/base.py"


def load_reference_base():
    """exec base.py under stubs for its two un-importable imports."""
    mmcv = types.ModuleType("mmcv")
    runner = types.ModuleType("mmcv.runner")
    runner.force_fp32 = lambda *a, **k: (lambda f: f)
    mmcv.runner = runner
    m3 = types.ModuleType("mmdet3d")
    ops = types.ModuleType("mmdet3d.ops")
    ops.bev_pool = None
    saved = {k: sys.modules.get(k) for k in ("mmcv", "mmcv.runner", "mmdet3d", "mmdet3d.ops")}
    sys.modules.update({"mmcv": mmcv, "mmcv.runner": runner, "mmdet3d": m3, "mmdet3d.ops": ops})
    try:
        mod = types.ModuleType("reference_vtransforms_base")
        exec(compile(open(REF_BASE).read(), REF_BASE, "exec"), mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def matrices(n_cam, B, seed):
    """Per-sample calibration + augmentation matrices [B, N, 4, 4] / [B, 4, 4] fp32: the synthetic rig of SURVEY.md §8d
    with train-time style image aug (resize 0.38-0.55, rotation +-5.4 deg, optional flip, crop) and LiDAR aug (rotation, scale,
    translation)."""
    rng = np.random.default_rng(seed)
    rig = synth.camera_rig(n_cam)
    rep = lambda a: np.repeat(np.asarray(a, np.float32)[None], B, 0)  # noqa: E731
    c2l = np.zeros((B, n_cam, 4, 4), np.float32)
    c2l[..., :3, :3], c2l[..., :3, 3], c2l[..., 3, 3] = rep(rig["camera2lidar_rots"]), rep(rig["camera2lidar_trans"]), 1
    K = np.zeros((B, n_cam, 4, 4), np.float32)
    K[..., :3, :3], K[..., 3, 3] = rep(rig["intrins"]), 1
    ia = np.tile(np.eye(4, dtype=np.float32), (B, n_cam, 1, 1))
    for b in range(B):
        for n in range(n_cam):
            s = rng.uniform(0.44, 0.52)
            a = np.radians(rng.uniform(-5.4, 5.4))
            R = s * np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
            t = np.array([-32.0 + rng.uniform(-8, 8), -176.0 + rng.uniform(-8, 8)])
            if rng.uniform() < 0.5:  # horizontal flip about the final image width
                F = np.array([[-1.0, 0.0], [0.0, 1.0]])
                R, t = F @ R, F @ t + np.array([704.0, 0.0])
            ia[b, n, :2, :2], ia[b, n, :2, 3] = R, t
    la = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    for b in range(B):
        a = rng.uniform(-0.4, 0.4)
        s = rng.uniform(0.9, 1.1)
        la[b, :3, :3] = s * np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        la[b, :3, 3] = rng.uniform(-0.5, 0.5, 3)
    l2c = np.linalg.inv(c2l.astype(np.float64))
    l2i = (K.astype(np.float64) @ l2c).astype(np.float32)
    return dict(c2l=c2l, K=K, ia=ia, la=la, l2i=l2i)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(ref, cfg, B, n_cam, seed, sweeps):
    class Capture(ref.BaseDepthTransform):
        def get_cam_feats(self, img, depth, mats):
            self.cap_depth = depth
            return torch.zeros(1)

        def bev_pool(self, geom_feats, x):
            # the reference's own prologue, base.py:149-169, up to the op call
            self.cap_geom = geom_feats.clone()
            Nprime = geom_feats.numel() // 3
            Bn = geom_feats.shape[0]
            g = ((geom_feats - (self.bx - self.dx / 2.0)) / self.dx).long().view(Nprime, 3)
            batch_ix = torch.cat([torch.full([Nprime // Bn, 1], ix, dtype=torch.long) for ix in range(Bn)])
            g = torch.cat((g, batch_ix), 1)
            kept = ((g[:, 0] >= 0) & (g[:, 0] < self.nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < self.nx[1]) & (g[:, 2] >= 0)
                    & (g[:, 2] < self.nx[2]))
            self.cap_cells, self.cap_kept = g, kept
            return None

    vt = Capture(256, 80, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"], cfg["zbound"],
                 cfg["dbound"], depth_input="scalar", height_expand=False, add_depth_features=False)
    m = matrices(n_cam, B, seed)
    pts = [synth.lidar_points(seed=seed * 10 + b, sweeps=sweeps) for b in range(B)]
    t = lambda a: torch.from_numpy(a.copy())  # noqa: E731
    img = torch.zeros(B, n_cam, 1, 1, 1)
    # forward(img, points, radar, sensor2ego, lidar2ego, lidar2camera, lidar2image, cam_intrinsic, camera2lidar,
    #         img_aug_matrix, lidar_aug_matrix, metas)   base.py:241-256
    # torch.inverse (LAPACK inside MKL) is layout-sensitive: the inverse of the strided [..., :3, :3] view of a 4x4 differs in
    # the last bit from the inverse of a contiguous copy.  Record the very matrices the reference's calls returned.
    mats4 = {k: t(v) for k, v in m.items()}
    calls = []
    real_inverse = torch.inverse

    def recording_inverse(x):
        # .contiguous(): same values, row-major layout.  LAPACK hands back column-major 3x3s, and MKL's sgemm picks its code
        # path by operand layout x thread count x n: with a column-major left operand `inverse(aug).matmul(points.T)`
        # (base.py:292) is a k-ascending FMA chain at 8 threads / n >= 60k but some other evaluation order at 1 thread or small n
        # (probed in this container) — not a property of the reference's code.  Row-major operands take the FMA-chain path
        # under every thread count and size probed, the same arithmetic as the two GEMMs that follow (:296, :305).
        r = real_inverse(x).contiguous()
        calls.append(r.clone())
        return r

    torch.inverse = recording_inverse
    try:
        # forward(img, points, radar, sensor2ego, lidar2ego, lidar2camera, lidar2image, cam_intrinsic, camera2lidar,
        #         img_aug_matrix, lidar_aug_matrix, metas)   base.py:241-256
        vt.forward(img, [t(p) for p in pts], None, mats4["c2l"], mats4["c2l"], None, mats4["l2i"], mats4["K"], mats4["c2l"],
                   mats4["ia"], mats4["la"], None)
    finally:
        torch.inverse = real_inverse
    assert len(calls) == B + 2     # B x inverse(lidar_aug[:3,:3]) in the raster loop, then post_rots and intrins (:106, :118)
    out = dict(m)
    out["frustum"] = vt.frustum.detach().numpy()
    out["dx"], out["bx"], out["nx"] = vt.dx.numpy(), vt.bx.numpy(), vt.nx.numpy()
    out["origin"] = (vt.bx - vt.dx / 2.0).detach().numpy()
    out["inv_lidar_aug_rot"] = torch.stack(calls[:B]).numpy()
    out["inv_post_rots"] = calls[B].numpy()
    out["inv_intrins"] = calls[B + 1].numpy()
    out["combine"] = mats4["c2l"][..., :3, :3].matmul(calls[B + 1]).numpy()      # base.py:118
    out["points_sha256"] = np.array([sha(p) for p in pts])
    out["points_seed"] = np.array([seed * 10 + b for b in range(B)])
    out["points_sweeps"] = np.array(sweeps)
    geom = vt.cap_geom.numpy()
    cells = vt.cap_cells.numpy()
    kept = vt.cap_kept.numpy()
    depth = vt.cap_depth.numpy()          # [B, N, 1, iH, iW]
    lin = np.flatnonzero(depth.reshape(-1))
    out["depth_shape"] = np.array(depth.shape)
    out["depth_lin"] = lin.astype(np.int32)
    out["depth_val"] = depth.reshape(-1)[lin]
    return out, geom, cells, kept


def main():
    torch.set_num_threads(1)
    ref = load_reference_base()
    fix = {}
    small = dict(synth.CL_CONFIG, feature_size=(8, 22), dbound=(1.0, 60.0, 2.0))
    out, geom, cells, kept = run_case(ref, small, B=2, n_cam=6, seed=3, sweeps=2)
    for k, v in out.items():
        fix["small_" + k] = v
    fix["small_geom"] = geom
    fix["small_cells"] = cells.astype(np.int16)       # |cell| < 2^15 here (checked)
    assert np.array_equal(fix["small_cells"].astype(np.int64), cells)
    fix["small_kept"] = kept
    out, geom, cells, kept = run_case(ref, synth.CL_CONFIG, B=1, n_cam=6, seed=5, sweeps=2)
    for k, v in out.items():
        fix["flag_" + k] = v
    fix["flag_geom_sha256"] = np.array(sha(geom))
    fix["flag_cells_sha256"] = np.array(sha(cells.astype(np.int32)))
    fix["flag_kept_sha256"] = np.array(sha(kept.astype(np.uint8)))
    fix["flag_n_kept"] = np.array(int(kept.sum()))
    path = os.path.join(HERE, "vtransform_ref.npz")
    np.savez_compressed(path, **fix)
    print(path, os.path.getsize(path) // 1024, "KiB;  small geom", fix["small_geom"].shape, "kept", int(fix["small_kept"].sum()),
          "depth hits", fix["small_depth_lin"].shape[0], "| flagship kept", int(fix["flag_n_kept"]), "depth hits",
          fix["flag_depth_lin"].shape[0])


if __name__ == "__main__":
    main()
