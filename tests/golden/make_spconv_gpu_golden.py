"""Generates tests/golden/spconv_gpu_ref.npz with the REFERENCE's own GPU sparse-convolution ops on device tensors:
`get_indice_pairs_3d` (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:27-141 -> indice_cuda.cu:66-98 + torch::_unique, :130),
`indice_conv_half` / `indice_conv_fp32` (spconv_ops.h:260-361) and `indice_conv_backward_half` (:363-456), run on an MI355X from
oracle/_ref/sparse_conv_ext/sparse_conv_ext.so (the reference's extension hipified from /root/reference at build time by
oracle/ref_build.py — sources never enter this repository).  VERDICT r4 item 4.

    gpurun -- python tests/golden/make_spconv_gpu_golden.py gpurun_out/golden
then copy gpurun_out/golden/spconv_gpu_ref.npz into tests/golden/.

Cases: the inputs of the five CPU goldens (spconv_ref_*.npz; everything stored) and two layers on the flagship grid — the voxels of
one synthetic LiDAR sweep on 1440 x 1440 x 41: SubM 3x3x3 16->16 and the strided 3x3x3 / 2 / pad 1 16->32 convolution that leaves
level 1.  The order of the pairs inside one kernel offset comes from atomicAdd slots in the reference (indice_cuda.cu): pairs are
stored SORTED per offset (small cases) or as a SHA-256 of the sorted arrays (flagship); output rows of the flagship layers are
stored for a seeded sample of rows plus a SHA-256 of the whole tensor.

Round 6: three more flagship-grid cases, `flag_subm32 / 64 / 128` — SubM 3x3x3 C -> C over the
SAME voxels with their rows in ascending linear index (b, x, y, z), the order the staged-rows (slab) kernels and the fused training
path need — so that the kernels the time is spent in are pinned to `indice_conv_half` / `indice_conv_backward_half` on the GPU, not
only to float64 and the reference's CPU functors.  A SubM output row IS its input row, so no row order is involved."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from bevfusion_amd import synth  # noqa: E402
from oracle import get_conv_output_size, ref_build  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
SMALL = ["subm3", "conv_s2", "conv_p110", "conv_out", "subm_dense"]


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def sorted_pairs(pairs, num):
    """[K, 2, N] with the valid pairs of every offset sorted by (input row, output row), -1 behind them"""
    out = np.full_like(pairs, -1)
    for k in range(pairs.shape[0]):
        n = int(num[k])
        p = pairs[k, :, :n]
        o = np.lexsort((p[1], p[0]))
        out[k, :, :n] = p[:, o]
    return out


def run_case(ext, dev, indices, B, shape, ks, st, pd, subm, feats, w, og):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out_shape = list(shape) if subm else get_conv_output_size(list(shape), list(ks), list(st), list(pd), [1, 1, 1])
    res = ext.get_indice_pairs_3d(t(indices), B, [int(v) for v in out_shape], [int(v) for v in shape], [int(v) for v in ks],
                                  [int(v) for v in st], [int(v) for v in pd], [1, 1, 1], [0, 0, 0], int(subm), 0)
    torch.cuda.synchronize()
    out_inds, pairs, num = res
    m = out_inds.shape[0]
    r = dict(out_indices=out_inds.cpu().numpy(), indice_num=num.cpu().numpy(), out_shape=np.array(out_shape))
    r["pairs_sorted"] = sorted_pairs(pairs.cpu().numpy(), r["indice_num"])
    f32 = ext.indice_conv_fp32(t(feats), t(w), pairs, num, m, 0, int(subm))
    f16 = ext.indice_conv_half(t(feats).half(), t(w).half(), pairs, num, m, 0, int(subm))
    gi, gw = ext.indice_conv_backward_half(t(feats).half(), t(w).half(), t(og(m)).half(), pairs, num, 0, int(subm))
    torch.cuda.synchronize()
    r.update(out_fp32=f32.cpu().numpy(), out_half=f16.cpu().numpy(), out_grad=og(m), in_grad_half=gi.cpu().numpy(),
             filter_grad_half=gw.cpu().numpy())
    return r


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    ext = ref_build.load_ref("sparse_conv_ext")
    dev = torch.device("cuda:0")
    out = {}
    # round 6: the fixture's existing entries are carried over untouched (`--regenerate` rebuilds them); only missing cases are run
    have = os.path.join(GOLDEN, "spconv_gpu_ref.npz")
    if os.path.exists(have) and "--regenerate" not in sys.argv:
        with np.load(have) as z0:
            out = {k: z0[k] for k in z0.files}
    for name in SMALL:
        if f"{name}.out_indices" in out:
            continue
        z = np.load(os.path.join(GOLDEN, f"spconv_ref_{name}.npz"))
        rng = np.random.default_rng(100)
        og = lambda m, z=z: (z["out_grad"] if z["out_grad"].shape[0] == m else None)
        r = run_case(ext, dev, z["indices"], int(z["batch_size"]), z["spatial_shape"], z["ksize"], z["stride"], z["padding"],
                     int(z["subm"]), z["features"], z["filters"], og)
        for k, v in r.items():
            out[f"{name}.{k}"] = v
        same_set = {tuple(x) for x in r["out_indices"]} == {tuple(x) for x in z["out_indices"]}
        print(name, "M", r["out_indices"].shape[0], "out set == cpu golden:", same_set, "num equal:",
              bool(np.array_equal(r["indice_num"], z["indice_num"])),
              "fp32 vs cpu golden (needs the CPU row order):", "same order" if np.array_equal(r["out_indices"], z["out_indices"]) else "other order",
              "half vs fp32 max", float(np.abs(r["out_half"].astype(np.float32) - r["out_fp32"]).max()))
    # flagship grid: one LiDAR sweep's voxels, rows in the voxelizer's first-appearance order
    import oracle

    cfg = synth.CL_CONFIG
    pts = synth.lidar_points(seed=0, sweeps=1)
    _, c, _ = oracle.hard_voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
    # the oracle's coors are in the reference-CPU layout; the encoder's indices are (batch, x, y, z) on sparse_shape
    # [1440, 1440, 41] (bevfusion.py:169-197 pads the batch index in front of what the voxelizer returns)
    xyz = c if c[:, 0].max() >= 41 else c[:, ::-1]
    indices = np.concatenate([np.zeros((c.shape[0], 1), np.int32), xyz.astype(np.int32)], 1)
    out["flag.indices_sha256"] = sha(indices)
    out["flag.num_voxels"] = np.int64(indices.shape[0])
    rng = np.random.default_rng(7)
    for tag, ks, st, pd, subm, cin, cout in (("flag_subm", (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, 16, 16),
                                             ("flag_conv", (3, 3, 3), (2, 2, 2), (1, 1, 1), 0, 16, 32)):
        if f"{tag}.out_indices" in out:
            rng.standard_normal((indices.shape[0], cin)); rng.standard_normal(ks + (cin, cout))     # keep the stream where it was
            continue
        feats = rng.standard_normal((indices.shape[0], cin)).astype(np.float16).astype(np.float32)     # fp16-representable
        w = (rng.standard_normal(ks + (cin, cout)) * 0.1).astype(np.float16).astype(np.float32)
        ogs = {}

        def og(m, ogs=ogs, cout=cout):
            if m not in ogs:
                ogs[m] = np.random.default_rng(9).standard_normal((m, cout)).astype(np.float16).astype(np.float32)
            return ogs[m]

        r = run_case(ext, dev, indices, 1, cfg["sparse_shape"], ks, st, pd, subm, feats, w, og)
        m = r["out_indices"].shape[0]
        rows = np.sort(np.random.default_rng(3).choice(m, size=min(2000, m), replace=False))
        out[f"{tag}.features_sha256"], out[f"{tag}.filters"] = sha(feats), w
        out[f"{tag}.out_indices"], out[f"{tag}.indice_num"], out[f"{tag}.out_shape"] = r["out_indices"], r["indice_num"], r["out_shape"]
        out[f"{tag}.pairs_sorted_sha256"] = sha(r["pairs_sorted"])
        out[f"{tag}.rows"] = rows
        out[f"{tag}.out_fp32_rows"], out[f"{tag}.out_half_rows"] = r["out_fp32"][rows], r["out_half"][rows]
        out[f"{tag}.out_fp32_sha256"], out[f"{tag}.out_half_sha256"] = sha(r["out_fp32"]), sha(r["out_half"])
        in_rows = np.sort(np.random.default_rng(4).choice(indices.shape[0], size=2000, replace=False))
        out[f"{tag}.in_rows"], out[f"{tag}.in_grad_half_rows"] = in_rows, r["in_grad_half"][in_rows]
        out[f"{tag}.filter_grad_half"] = r["filter_grad_half"]
        print(tag, "N", indices.shape[0], "M", m, "pairs", int(r["indice_num"].sum()),
              "half vs fp32 max", float(np.abs(r["out_half"].astype(np.float32) - r["out_fp32"]).max()))
    # ---- round 6: wide SubM layers over the same voxels in linear order
    # (a full multi-sweep cloud at the inference cap: ~19 of 27 neighbours per row, like the bench's frames; the single sweep above
    # has 3.5)
    shape = cfg["sparse_shape"]
    _, c2, _ = oracle.hard_voxelize(synth.lidar_points(seed=1), cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"],
                                    cfg["max_voxels"][1])
    xyz2 = c2 if c2[:, 0].max() >= 41 else c2[:, ::-1]
    ind2 = np.concatenate([np.zeros((c2.shape[0], 1), np.int32), xyz2.astype(np.int32)], 1)
    key = (ind2[:, 1].astype(np.int64) * shape[1] + ind2[:, 2]) * shape[2] + ind2[:, 3]
    lin = np.ascontiguousarray(ind2[np.argsort(key, kind="stable")])
    out["flaglin.num_voxels"] = np.int64(lin.shape[0])
    out["flaglin.indices_sha256"] = sha(lin)
    for C in (32, 64, 128):
        tag = f"flag_subm{C}"
        if f"{tag}.rows" in out:
            continue
        rng = np.random.default_rng(700 + C)
        feats = (rng.standard_normal((lin.shape[0], C)) * 0.5).astype(np.float16).astype(np.float32)
        w = (rng.standard_normal((3, 3, 3, C, C)) * (0.6 / np.sqrt(27 * C))).astype(np.float16).astype(np.float32)
        ogv = (np.random.default_rng(900 + C).standard_normal((lin.shape[0], C)) * 0.5).astype(np.float16).astype(np.float32)
        r = run_case(ext, dev, lin, 1, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1, feats, w, lambda m, ogv=ogv: ogv)
        assert np.array_equal(r["out_indices"], lin)
        rows = np.sort(np.random.default_rng(3).choice(lin.shape[0], size=512, replace=False))
        out[f"{tag}.features_sha256"], out[f"{tag}.filters_sha256"], out[f"{tag}.out_grad_sha256"] = sha(feats), sha(w), sha(ogv)
        out[f"{tag}.indice_num"] = r["indice_num"]
        out[f"{tag}.rows"] = rows
        out[f"{tag}.out_fp32_rows"], out[f"{tag}.out_half_rows"] = r["out_fp32"][rows], r["out_half"][rows]
        out[f"{tag}.out_half_sha256"] = sha(r["out_half"])
        out[f"{tag}.in_grad_half_rows"] = r["in_grad_half"][rows]
        out[f"{tag}.filter_grad_half"] = r["filter_grad_half"]
        print(tag, "N", lin.shape[0], "pairs", int(r["indice_num"].sum()), "half vs fp32 max",
              float(np.abs(r["out_half"].astype(np.float32) - r["out_fp32"]).max()), "|out| max", float(np.abs(r["out_fp32"]).max()),
              "|wgrad| max", float(np.abs(r["filter_grad_half"].astype(np.float32)).max()))
    path = os.path.join(out_dir, "spconv_gpu_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/golden")
