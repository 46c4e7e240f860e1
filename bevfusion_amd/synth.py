"""Seeded synthetic nuScenes-shaped inputs (no dataset, no network): the workload generator used
by tests and bench.py.  Shapes and constants follow SURVEY.md §8d:

  * camera rig: 6 cameras, 1600x900 sensor, fx=fy=1266, cx=816, cy=491, yaw 0/∓55/∓110/180 deg,
    mounted (1.5, 0, -0.3) m from the LiDAR origin; eval-time image aug = resize 0.48, crop
    (-32, -176)  (datasets/pipelines/transforms_3d.py:123-131 with resize_lim [0.48, 0.48]);
  * frustum / geometry math restated from vtransforms/base.py:66-89, 92-135 in numpy fp32;
  * LiDAR: 32-beam spinning model, 10 sweeps, ~300k points after range crop, shuffled.

numpy only; everything is deterministic given the seed.
"""
import math

import numpy as np

# C+L flagship config (configs/nuscenes/det/transfusion/secfpn/camera+lidar/swint_v0p075)
CL_CONFIG = dict(
    image_size=(256, 704),
    feature_size=(32, 88),
    xbound=(-54.0, 54.0, 0.3),
    ybound=(-54.0, 54.0, 0.3),
    zbound=(-10.0, 10.0, 20.0),
    dbound=(1.0, 60.0, 0.5),
    channels=80,
    num_cameras=6,
    point_cloud_range=(-54.0, -54.0, -5.0, 54.0, 54.0, 3.0),
    voxel_size=(0.075, 0.075, 0.2),
    max_num_points=10,
    max_voxels=(120000, 160000),
    sparse_shape=(1440, 1440, 41),
    point_dim=5,
)

# BASELINE config 1: camera-only LSS, 1 camera, 64x64 BEV
LSS_SMALL_CONFIG = dict(
    image_size=(256, 704),
    feature_size=(32, 88),
    xbound=(-51.2, 51.2, 1.6),
    ybound=(-51.2, 51.2, 1.6),
    zbound=(-10.0, 10.0, 20.0),
    dbound=(1.0, 60.0, 1.0),
    channels=80,
    num_cameras=1,
)


def gen_dx_bx(xbound, ybound, zbound):
    """vtransforms/base.py:15-21 (fp32 tensors; nx by float division then int truncation)."""
    rows = [xbound, ybound, zbound]
    dx = np.array([r[2] for r in rows], dtype=np.float32)
    bx = np.array([r[0] + r[2] / 2.0 for r in rows], dtype=np.float32)
    nx = np.array([int((r[1] - r[0]) / r[2]) for r in rows], dtype=np.int64)
    return dx, bx, nx


def create_frustum(image_size, feature_size, dbound):
    """vtransforms/base.py:66-89 -> [D, fH, fW, 3] fp32 (u, v, depth)."""
    iH, iW = image_size
    fH, fW = feature_size
    import torch  # the reference builds these with torch.arange / torch.linspace; numpy's differ by 1 ulp

    ds = torch.arange(*dbound, dtype=torch.float).numpy()
    D = ds.shape[0]
    xs = torch.linspace(0, iW - 1, fW, dtype=torch.float).numpy()
    ys = torch.linspace(0, iH - 1, fH, dtype=torch.float).numpy()
    fr = np.empty((D, fH, fW, 3), dtype=np.float32)
    fr[..., 0] = xs[None, None, :]
    fr[..., 1] = ys[None, :, None]
    fr[..., 2] = ds[:, None, None]
    return fr


def camera_rig(num_cameras=6):
    """Returns dict of fp32 arrays: camera2lidar_rots [N,3,3], camera2lidar_trans [N,3],
    intrins [N,3,3], post_rots [N,3,3], post_trans [N,3]."""
    yaws = [0.0, -55.0, 55.0, -110.0, 110.0, 180.0][:num_cameras]
    base = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)  # cam(x right,y down,z fwd) -> lidar
    rots, trans, intr, prot, ptr = [], [], [], [], []
    for y in yaws:
        a = math.radians(y)
        rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]], dtype=np.float64)
        rots.append(rz @ base)
        trans.append(rz @ np.array([1.5, 0.0, -0.3]))
        intr.append(np.array([[1266.0, 0, 816.0], [0, 1266.0, 491.0], [0, 0, 1]]))
        prot.append(np.diag([0.48, 0.48, 1.0]))
        ptr.append(np.array([-32.0, -176.0, 0.0]))
    f = lambda l: np.stack(l).astype(np.float32)
    return dict(camera2lidar_rots=f(rots), camera2lidar_trans=f(trans), intrins=f(intr), post_rots=f(prot),
                post_trans=f(ptr))


def get_geometry(frustum, rig, batch=1):
    """vtransforms/base.py:92-135 restated in numpy fp32 -> [B, N, D, fH, fW, 3] lidar-frame points."""
    N = rig["intrins"].shape[0]
    out = np.empty((batch, N) + frustum.shape, dtype=np.float32)
    for n in range(N):
        pts = frustum - rig["post_trans"][n].reshape(1, 1, 1, 3)
        inv_post = np.linalg.inv(rig["post_rots"][n].astype(np.float32)).astype(np.float32)
        pts = np.einsum("ij,dhwj->dhwi", inv_post, pts).astype(np.float32)
        pts = np.concatenate([pts[..., :2] * pts[..., 2:3], pts[..., 2:3]], axis=-1).astype(np.float32)
        combine = (rig["camera2lidar_rots"][n] @ np.linalg.inv(rig["intrins"][n]).astype(np.float32)).astype(np.float32)
        pts = np.einsum("ij,dhwj->dhwi", combine, pts).astype(np.float32)
        pts = pts + rig["camera2lidar_trans"][n].reshape(1, 1, 1, 3)
        out[:, n] = pts[None]
    return out


def bev_pool_inputs(cfg=CL_CONFIG, batch=1, channels=None, seed=0, with_feats=True, feat_dtype=np.float32):
    """Everything `BaseTransform.bev_pool` consumes for one batch: unfiltered geometry [N',3],
    features [N',C], and the grid (origin = bx - dx/2, dx, nx).  N' = B*Ncam*D*fH*fW."""
    dx, bx, nx = gen_dx_bx(cfg["xbound"], cfg["ybound"], cfg["zbound"])
    fr = create_frustum(cfg["image_size"], cfg["feature_size"], cfg["dbound"])
    rig = camera_rig(cfg["num_cameras"])
    geom = get_geometry(fr, rig, batch=batch).reshape(-1, 3)
    c = channels or cfg["channels"]
    out = dict(geom=geom, dx=dx, bx=bx, nx=nx, origin=(bx - dx / np.float32(2.0)).astype(np.float32), channels=c,
               batch=batch, D=fr.shape[0])
    if with_feats:
        rng = np.random.default_rng(seed)
        out["feats"] = rng.standard_normal((geom.shape[0], c), dtype=np.float32).astype(feat_dtype)
    return out


def lidar_points(seed=0, sweeps=10, cfg=CL_CONFIG):
    """32-beam x 1084-azimuth spinning LiDAR, `sweeps` sweeps with ego motion, ground + walls,
    cropped to the point-cloud range and SHUFFLED.  -> [~300k, 5] fp32 (x, y, z, intensity, dt)."""
    rng = np.random.default_rng(seed)
    elev = np.radians(np.linspace(-30.67, 10.67, 32))
    azim = np.linspace(-np.pi, np.pi, 1084, endpoint=False)
    sector_wall = rng.uniform(8.0, 50.0, size=72)  # one wall distance per 5 degree sector
    pts = []
    for s in range(sweeps):
        ego = np.array([0.5 * s, 0.02 * s, 0.0])
        el, az = np.meshgrid(elev, azim, indexing="ij")
        dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1).reshape(-1, 3)
        # range to ground plane z = -1.84 (sensor at z=0), or to the sector wall
        with np.errstate(divide="ignore", invalid="ignore"):
            t_ground = np.where(dirs[:, 2] < -1e-3, -1.84 / dirs[:, 2], np.inf)
        sector = ((np.degrees(np.arctan2(dirs[:, 1], dirs[:, 0])) + 180.0) // 5.0).astype(np.int64) % 72
        horiz = np.maximum(np.hypot(dirs[:, 0], dirs[:, 1]), 1e-6)
        t_wall = sector_wall[sector] / horiz
        t = np.minimum(t_ground, t_wall)
        t = t * (1.0 + 0.003 * rng.standard_normal(t.shape))
        p = dirs * t[:, None] - ego[None]
        p[:, 2] += 0.02 * rng.standard_normal(p.shape[0])
        inten = rng.uniform(0, 255, size=(p.shape[0], 1))
        dt = np.full((p.shape[0], 1), 0.05 * s)
        pts.append(np.concatenate([p, inten, dt], 1))
    pts = np.concatenate(pts, 0).astype(np.float32)
    r = cfg["point_cloud_range"]
    m = ((pts[:, 0] >= r[0]) & (pts[:, 0] < r[3]) & (pts[:, 1] >= r[1]) & (pts[:, 1] < r[4]) & (pts[:, 2] >= r[2])
         & (pts[:, 2] < r[5]))
    pts = pts[m]
    rng.shuffle(pts, axis=0)
    return np.ascontiguousarray(pts)


def rigged_geometry(B, n_cam, D, fh, fw, seed, pitch_deg=1.5, roll_deg=1.0, rot_deg=5.4, flip=True):
    """[B, n_cam, D, fh, fw, 3] lidar-frame frustum points (float64 restatement of base.py:92-135) of a rig whose cameras are
    pitched / rolled a little (as mounted cameras are) and whose image augmentation rotates and flips (training-time
    augmentation, transforms_3d.py:85-118): a column's rows then cross a few BEV cells."""
    rng = np.random.default_rng(seed)
    cfg = CL_CONFIG
    iH, iW = cfg["image_size"]
    rig = camera_rig(n_cam)
    ds = np.arange(1.0, 1.0 + 0.5 * D, 0.5)[:D]
    xs, ys = np.linspace(0, iW - 1, fw), np.linspace(0, iH - 1, fh)
    fr = np.stack(np.broadcast_arrays(xs[None, None, :], ys[None, :, None], ds[:, None, None]), -1)     # [D, fh, fw, 3]
    out = np.empty((B, n_cam, D, fh, fw, 3))
    for b in range(B):
        for n in range(n_cam):
            a = math.radians(rng.uniform(-rot_deg, rot_deg))
            s = 0.48 * rng.uniform(0.9, 1.1)
            post_rot = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]]) * np.array([s, s, 1.0])
            if flip and rng.random() < 0.5:
                post_rot = np.diag([-1.0, 1.0, 1.0]) @ post_rot
            post_tran = np.array([rng.uniform(-40, 0) + (iW if post_rot[0, 0] < 0 else 0), rng.uniform(-190, -160), 0.0])
            pr, rr = math.radians(rng.uniform(-pitch_deg, pitch_deg)), math.radians(rng.uniform(-roll_deg, roll_deg))
            rx = np.array([[1, 0, 0], [0, math.cos(pr), -math.sin(pr)], [0, math.sin(pr), math.cos(pr)]])
            rz = np.array([[math.cos(rr), -math.sin(rr), 0], [math.sin(rr), math.cos(rr), 0], [0, 0, 1]])
            c2l = rig["camera2lidar_rots"][n].astype(np.float64) @ rx @ rz
            pts = (fr - post_tran) @ np.linalg.inv(post_rot).T
            pts = np.concatenate([pts[..., :2] * pts[..., 2:3], pts[..., 2:3]], -1)
            pts = pts @ (c2l @ np.linalg.inv(rig["intrins"][n].astype(np.float64))).T + rig["camera2lidar_trans"][n]
            out[b, n] = pts
    return out.astype(np.float32)


def training_augmentation(rng, B, n_cam, cfg=CL_CONFIG, ori_shape=(1600, 900), resize_lim=(0.38, 0.55), bot_pct_lim=(0.0, 0.0),
                          rot_lim_deg=(-5.4, 5.4), rand_flip=True, lidar_scale=(0.9, 1.1), lidar_rot=(-0.78539816, 0.78539816),
                          lidar_trans_std=0.5):
    """One training batch's augmentation matrices, drawn the way the reference's pipeline draws them per sample and per camera:
    ImageAug3D (datasets/pipelines/transforms_3d.py:85-131 sampling, :134-165 the post-homography: resize, crop, flip, rotate about the
    crop centre; configs/nuscenes/default.yaml:13-15 resize [0.38, 0.55], rotate +-5.4 deg; bot_pct_lim [0, 0], rand_flip) and
    GlobalRotScaleTrans (:196-230; default.yaml:20-23 scale 0.9-1.1, rotate +-pi/4, translate N(0, 0.5) per axis).
    -> dict(post_rots [B,N,3,3], post_trans [B,N,3], extra_rots [B,3,3], extra_trans [B,3]) fp32: what BaseTransform._split_mats
    hands get_geometry (img_aug_matrix[..., :3, :3] / [..., :3, 3], lidar_aug_matrix likewise)."""
    W, H = ori_shape
    fH, fW = cfg["image_size"]
    post_rots = np.tile(np.eye(3, dtype=np.float64), (B, n_cam, 1, 1))
    post_trans = np.zeros((B, n_cam, 3))
    for b in range(B):
        for n in range(n_cam):
            resize = rng.uniform(*resize_lim)
            newW, newH = int(W * resize), int(H * resize)
            crop_h = int((1 - rng.uniform(*bot_pct_lim)) * newH) - fH
            crop_w = int(rng.uniform(0, max(0, newW - fW)))
            flip = bool(rand_flip and rng.integers(0, 2))
            theta = rng.uniform(*rot_lim_deg) / 180.0 * math.pi
            rot = np.eye(2) * resize
            tr = -np.array([crop_w, crop_h], dtype=np.float64)
            if flip:
                A = np.array([[-1.0, 0.0], [0.0, 1.0]])
                rot, tr = A @ rot, A @ tr + np.array([fW, 0.0])
            A = np.array([[math.cos(theta), math.sin(theta)], [-math.sin(theta), math.cos(theta)]])
            c = np.array([fW, fH], dtype=np.float64) / 2
            rot, tr = A @ rot, A @ tr + (A @ (-c) + c)
            post_rots[b, n, :2, :2] = rot
            post_trans[b, n, :2] = tr
    extra_rots = np.empty((B, 3, 3))
    extra_trans = np.empty((B, 3))
    for b in range(B):
        scale, th = rng.uniform(*lidar_scale), rng.uniform(*lidar_rot)
        t = rng.normal(0.0, lidar_trans_std, size=3)
        R = np.array([[math.cos(th), -math.sin(th), 0.0], [math.sin(th), math.cos(th), 0.0], [0.0, 0.0, 1.0]])
        extra_rots[b] = R.T * scale
        extra_trans[b] = t * scale
    f = lambda a: a.astype(np.float32)
    return dict(post_rots=f(post_rots), post_trans=f(post_trans), extra_rots=f(extra_rots), extra_trans=f(extra_trans))
