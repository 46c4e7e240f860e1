"""CPU: host-side mirrors of the reference interface — configs load unchanged, registries resolve the
reference's type names, module trees / state-dict keys / geometry formulas match."""
import glob
import os

import numpy as np
import pytest
import torch

from bevfusion_amd import synth
from bevfusion_amd.config import build_hot_path, load_config, recursive_eval
from bevfusion_amd.registry import BACKBONES, CONV_LAYERS, VTRANSFORMS, build_conv_layer, build_norm_layer
from bevfusion_amd.sparse_encoder import SparseEncoder
from bevfusion_amd.vtransforms import DepthLSSTransform, LSSTransform, gen_dx_bx

REF_CONFIGS = "/root/reference/configs"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="/root/reference not present (GPU box)")


@needs_ref
def test_every_reference_config_loads():
    files = sorted(glob.glob(REF_CONFIGS + "/**/*.yaml", recursive=True))
    assert len(files) >= 20
    for f in files:
        cfg = load_config(f)
        assert isinstance(cfg, dict)
        assert "${" not in repr(cfg), f           # every expression evaluated


@needs_ref
def test_flagship_config_builds_the_hot_path_modules():
    cfg = load_config(REF_CONFIGS + "/nuscenes/det/transfusion/secfpn/camera+lidar/swint_v0p075/convfuser.yaml")
    assert cfg["model"]["encoders"]["camera"]["vtransform"]["feature_size"] == [32, 88]   # ${[image_size[0]//8, ...]}
    hp = build_hot_path(cfg)
    vt = hp["vtransform"]
    assert isinstance(vt, DepthLSSTransform) and vt.D == 118 and vt.C == 80 and vt.nx.tolist() == [360, 360, 1]
    assert hp["voxelize"].max_voxels == (120000, 160000) and hp["voxelize"].grid_size.tolist() == [1440, 1440, 40]
    enc = hp["lidar_backbone"]
    assert isinstance(enc, SparseEncoder) and list(enc.sparse_shape) == [1440, 1440, 41]
    w = enc.state_dict()
    assert tuple(w["conv_input.0.weight"].shape) == (3, 3, 3, 5, 16)            # [kx,ky,kz,Cin,Cout] (conv.py:100)
    assert tuple(w["conv_out.0.weight"].shape) == (1, 1, 3, 128, 128)
    assert "encoder_layers.encoder_layer1.0.conv1.weight" in w and "encoder_layers.encoder_layer1.0.bn2.running_var" in w
    # camera-only LSS config (BASELINE config 1 family)
    cfg2 = load_config(REF_CONFIGS + "/nuscenes/det/centerhead/lssfpn/camera/256x704/swint/default.yaml")
    vt2 = build_hot_path(cfg2)["vtransform"]
    assert isinstance(vt2, LSSTransform)


def test_recursive_eval_semantics():
    cfg = {"a": [256, 704], "b": "${[a[0] // 8, a[1] // 8]}", "c": {"d": "${b}", "e": "${a[0] * 2}"}, "s": "${'x' + 'y'}"}
    from bevfusion_amd.config import _attrify, _plain

    out = _plain(recursive_eval(_attrify(cfg)))
    assert out["b"] == [32, 88] and out["c"]["d"] == [32, 88] and out["c"]["e"] == 512 and out["s"] == "xy"


def test_registries_resolve_reference_type_names():
    assert "SubMConv3d" in CONV_LAYERS and "SparseConv3d" in CONV_LAYERS
    assert "SparseEncoder" in BACKBONES and "DepthLSSTransform" in VTRANSFORMS and "LSSTransform" in VTRANSFORMS
    conv = build_conv_layer(dict(type="SubMConv3d", indice_key="subm1"), 5, 16, 3, stride=1, padding=1, bias=False)
    assert conv.subm and conv.indice_key == "subm1" and conv.bias is None and conv.padding == [1, 1, 1]
    name, bn = build_norm_layer(dict(type="BN1d", eps=1e-3, momentum=0.01), 16)
    assert name == "bn" and isinstance(bn, torch.nn.BatchNorm1d) and bn.eps == 1e-3 and bn.momentum == 0.01


def test_sparse_encoder_tree_matches_reference_layout():
    cfg = synth.CL_CONFIG
    enc = SparseEncoder(5, list(cfg["sparse_shape"]), order=["conv", "norm", "act"], output_channels=128,
                        encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                        encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock")
    from bevfusion_amd import spconv

    convs = [m for m in enc.modules() if isinstance(m, spconv.SparseConvolution)]
    assert len(convs) == 21 and sum(c.subm for c in convs) == 17          # 17 SubM + 4 strided
    assert enc.encoder_layers.encoder_layer3[2][0].padding == [1, 1, 0]
    assert enc.conv_out[0].kernel_size == [1, 1, 3] and enc.conv_out[0].stride == [1, 1, 2]
    assert enc.conv_input[0].indice_key == "subm1" and enc.encoder_layers.encoder_layer1[0].conv1.indice_key is None


def test_vtransform_geometry_matches_numpy_restatement():
    """torch `get_geometry` (module) == numpy restatement used by the synthetic rig (both follow base.py:92-135)."""
    cfg = synth.LSS_SMALL_CONFIG
    vt = LSSTransform(256, 80, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"], cfg["zbound"],
                      cfg["dbound"])
    dx, bx, nx = gen_dx_bx(cfg["xbound"], cfg["ybound"], cfg["zbound"])
    sdx, sbx, snx = synth.gen_dx_bx(cfg["xbound"], cfg["ybound"], cfg["zbound"])
    assert np.array_equal(dx.numpy(), sdx) and np.array_equal(bx.numpy(), sbx) and nx.tolist() == snx.tolist()
    fr = synth.create_frustum(cfg["image_size"], cfg["feature_size"], cfg["dbound"])
    assert np.array_equal(vt.frustum.numpy(), fr) and vt.D == fr.shape[0] == 59
    rig = synth.camera_rig(3)
    t = lambda a: torch.from_numpy(a)[None]
    geom = vt.get_geometry(t(rig["camera2lidar_rots"]), t(rig["camera2lidar_trans"]), t(rig["intrins"]),
                           t(rig["post_rots"]), t(rig["post_trans"]))
    ref = synth.get_geometry(fr, rig, batch=1)
    assert geom.shape == ref.shape
    assert np.max(np.abs(geom.numpy() - ref)) < 2e-3                       # metres; fp32 matmul order differs


def test_product_has_no_oracle_import():
    """The oracle is test infrastructure: nothing under bevfusion_amd/ may import it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bevfusion_amd")
    for f in glob.glob(root + "/**/*.py", recursive=True):
        src = open(f).read()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_oracle_vtransform_restatements_agree_with_module_and_rig():
    """oracle.lss_geometry == the module's broadcasting formulation (torch CPU) == synth.get_geometry up to the 3x3
    inverse (the oracle's stand-in is float64-rounded, torch's is fp32 LAPACK; the bit-exact pin with the reference's own
    inverses is tests/test_oracle_vtransform.py); oracle.depth_raster == the module's torch formulation on inputs without
    pixel collisions."""
    import oracle

    cfg = synth.CL_CONFIG
    vt = DepthLSSTransform(256, 80, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"], cfg["zbound"],
                           cfg["dbound"], downsample=2)
    rig = synth.camera_rig(6)
    t = lambda a: torch.from_numpy(a)[None]
    geom = vt.get_geometry(t(rig["camera2lidar_rots"]), t(rig["camera2lidar_trans"]), t(rig["intrins"]),
                           t(rig["post_rots"]), t(rig["post_trans"]))
    ref = oracle.lss_geometry(vt.frustum.numpy(), rig["post_rots"][None], rig["post_trans"][None],
                              rig["camera2lidar_rots"][None], rig["camera2lidar_trans"][None], rig["intrins"][None])
    assert np.max(np.abs(geom.numpy() - ref)) < 2e-3
    # depth raster: a sparse, collision-free set of points
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.uniform(-40, 40, (400, 2)), rng.uniform(-2, 1, (400, 1)), rng.random((400, 2))], 1).astype(np.float32)
    c2l = np.zeros((6, 4, 4), np.float32)
    c2l[:, :3, :3], c2l[:, :3, 3], c2l[:, 3, 3] = rig["camera2lidar_rots"], rig["camera2lidar_trans"], 1
    K = np.zeros((6, 4, 4), np.float32)
    K[:, :3, :3], K[:, 3, 3] = rig["intrins"], 1
    ia = np.zeros((6, 4, 4), np.float32)
    ia[:, :3, :3], ia[:, :3, 3], ia[:, 3, 3] = rig["post_rots"], rig["post_trans"], 1
    l2i = (K.astype(np.float64) @ np.linalg.inv(c2l.astype(np.float64))).astype(np.float32)
    la = np.eye(4, dtype=np.float32)
    ref_d, winner = oracle.depth_raster(pts, l2i, ia, la, cfg["image_size"])
    got = vt.depth_raster(torch.zeros(1, 6, 1, 1, 1), [torch.from_numpy(pts)], torch.from_numpy(l2i)[None],
                          torch.from_numpy(ia)[None], torch.from_numpy(la)[None])
    assert int((ref_d > 0).sum()) > 20
    hit = winner >= 0
    assert np.array_equal(got[0, :, 0].numpy() > 0, hit)
    assert np.max(np.abs(got[0, :, 0].numpy() - ref_d[:, 0])) < 1e-3


def test_factored_cam_feats_is_the_reference_outer_product():
    """FactoredCamFeats.materialize() == depth_lss.py:92-97 (outer product, view, permute), on CPU."""
    from bevfusion_amd.vtransforms import FactoredCamFeats

    B, N, D, C, fH, fW = 2, 3, 5, 4, 2, 3
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * N, D + C, fH, fW, generator=g)
    depth = x[:, :D].softmax(dim=1)
    ref = (depth.unsqueeze(1) * x[:, D:D + C].unsqueeze(2)).view(B, N, C, D, fH, fW).permute(0, 1, 3, 4, 5, 2)
    fz = FactoredCamFeats(depth.view(B, N, D, fH, fW), x[:, D:D + C].view(B, N, C, fH, fW))
    assert torch.equal(fz.materialize(), ref)


def test_fused_encoder_path_is_gated_on_cpu_and_keeps_module_semantics():
    """On CPU tensors / fp32 / grad-enabled the fused path must not engage (the module path then raises loudly because
    the HIP extension has no CPU path)."""
    import pytest

    from bevfusion_amd.sparse_encoder import SparseEncoder
    from bevfusion_amd.spconv import fused

    enc = SparseEncoder(5, [16, 16, 9], order=["conv", "norm", "act"]).eval()
    x, c = torch.randn(10, 5), torch.zeros(10, 4, dtype=torch.int32)
    with torch.no_grad():
        assert not fused.encoder_supported(enc, x)
        with pytest.raises(RuntimeError, match="GPU tensor"):
            enc(x, c, 1)
    assert fused.INDEX_HASH == 0 and fused.INDEX_RANK == 1


def test_spconv_family_is_registered_with_reference_signatures():
    """The ten classes the reference registers in CONV_LAYERS (spconv/conv.py:226-455) build through build_conv_layer with
    its keyword set, keep its parameter layout ([k..., Cin, Cout]) and flags; the pooling and scatter modules construct."""
    from bevfusion_amd import spconv
    from bevfusion_amd.voxel import DynamicScatter

    names = ["SparseConv2d", "SparseConv3d", "SparseConv4d", "SparseConvTranspose2d", "SparseConvTranspose3d",
             "SparseInverseConv2d", "SparseInverseConv3d", "SubMConv2d", "SubMConv3d", "SubMConv4d"]
    for n in names:
        assert n in CONV_LAYERS and hasattr(spconv, n)
    up = build_conv_layer(dict(type="SparseConvTranspose3d", indice_key="up1"), 8, 4, 3, stride=2, padding=1, bias=False)
    assert up.transposed and not up.subm and not up.inverse and tuple(up.weight.shape) == (3, 3, 3, 8, 4)
    inv = build_conv_layer(dict(type="SparseInverseConv3d", indice_key="down1"), 8, 4, 3, bias=False)
    assert inv.inverse and inv.indice_key == "down1" and inv.stride == [1, 1, 1]
    c2 = build_conv_layer(dict(type="SubMConv2d"), 4, 6, (3, 5), bias=True)
    assert c2.ndim == 2 and c2.subm and tuple(c2.weight.shape) == (3, 5, 4, 6) and tuple(c2.bias.shape) == (6,)
    c4 = spconv.SparseConv4d(2, 2, 3)
    assert c4.ndim == 4 and tuple(c4.weight.shape) == (3, 3, 3, 3, 2, 2)
    with pytest.raises(AssertionError):                       # dilation and stride both != 1: conv.py:85
        spconv.SparseConv3d(2, 2, 3, stride=2, dilation=2)
    pool = spconv.SparseMaxPool3d(3, 2, 1)
    assert pool.kernel_size == [3, 3, 3] and pool.stride == [2, 2, 2] and pool.padding == [1, 1, 1] and not pool.subm
    assert spconv.SparseMaxPool2d((2, 3)).kernel_size == [2, 3]
    ds = DynamicScatter([0.1, 0.1, 0.2], [0, 0, 0, 1, 1, 1], True)
    assert ds.average_points and "average_points=True" in repr(ds)
    assert spconv.get_deconv_output_size([5, 6], [3, 3], [2, 2], [1, 1], [1, 1], [1, 0]) == [10, 11]


def test_lift_to_3d_keeps_offsets_and_order():
    from bevfusion_amd.spconv.ops import _lift_to_3d

    ind = torch.tensor([[0, 3, 4], [1, 0, 2]], dtype=torch.int32)
    g = dict(shape=[8, 9], out_shape=[4, 5], ksize=[3, 2], stride=[2, 2], padding=[1, 0], dilation=[1, 1])
    ind3, g3 = _lift_to_3d(ind, g, 2)
    assert ind3.tolist() == [[0, 3, 4, 0], [1, 0, 2, 0]]
    assert g3 == dict(shape=[8, 9, 1], out_shape=[4, 5, 1], ksize=[3, 2, 1], stride=[2, 2, 1], padding=[1, 0, 0],
                      dilation=[1, 1, 1])
    same, gs = _lift_to_3d(ind3, g3, 3)
    assert same is ind3 and gs is g3
