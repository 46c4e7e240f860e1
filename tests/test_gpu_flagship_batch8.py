"""GPU parity at the BENCHMARKED configuration — BASELINE.json configs[3]: the C+L flagship sizes at 8 frames per step — for
exactly the kernels bench.py times there and no smaller test reaches (VERDICT r1 "missing" #2):

  (i)   bev_pool over 8 frames in one launch: `bev_pool_fwd_cells_vec_kernel<float4,4,4>` (selected only above 200 000 cells)
        and the fused depth (x) context kernel on the same plan, each frame against the float64 oracle, <= 1e-4;
  (ii)  `voxelize_batch_device` on 8 point clouds spread over 4 HIP streams, eagerly and replayed from a captured HIP graph:
        coordinates, per-voxel counts and the packed order bit-exact against `oracle.voxelize_batch`;
  (iii) the fused SparseEncoder at 8 frames (batched gather tilings of `fused._BATCHED_VARIANTS` + the slab kernels) against the
        module-by-module path, and its level chain (active sets, row order, counts) against the oracle;
  (iv)  one flagship frame, stage by stage: the fused path's fp16 output of a SubM layer and of the strided convolution of every
        level against `oracle.indice_conv` (float64) fed with the GPU's own stage input, <= 2e-3 * (1 + max|ref|).

References: bevfusion.py:169-197 (voxelize), sparse_encoder.py:100-132, base.py:141-176, bev_pool_cuda.cu:20-42."""
import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import synth
from bevfusion_amd.bev_pool import BevPoolPlan
from bevfusion_amd.sparse_encoder import SparseEncoder
from bevfusion_amd.spconv import fused
from bevfusion_amd.voxel import voxelize_batch_device

from conftest import record_parity

pytestmark = pytest.mark.gpu
CFG = synth.CL_CONFIG
B8 = 8


def flagship_encoder(dev, dtype=torch.float16, seed=0):
    torch.manual_seed(seed)
    enc = SparseEncoder(5, list(CFG["sparse_shape"]), order=["conv", "norm", "act"], output_channels=128,
                        encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                        encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock")
    for m in enc.modules():                       # non-trivial BatchNorm statistics (random-init defaults are the identity)
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.7, 1.3)
            m.weight.data.uniform_(0.7, 1.3)
            m.bias.data.normal_(0, 0.1)
    return enc.to(dev).to(dtype).eval()


def test_bev_pool_eight_frames_one_launch_vs_float64_oracle(dev):
    inp = synth.bev_pool_inputs(CFG, batch=B8, with_feats=False)
    H, W, D = (int(v) for v in inp["nx"])
    C = 80
    assert B8 * D * H * W > 200000                                    # the size class that selects <float4, 4, 4>
    geom = torch.from_numpy(inp["geom"]).to(dev)
    n_frame = geom.shape[0] // B8
    plan = BevPoolPlan.from_geometry(geom, B8, inp["origin"], inp["dx"], inp["nx"])
    coords1, kept1 = oracle.bev_cell_index(inp["geom"][:n_frame], 1, inp["origin"], inp["dx"], inp["nx"])   # same rig per frame
    assert plan.n_kept() == B8 * int(kept1.sum())
    gen = torch.Generator(device=dev).manual_seed(11)
    feats = torch.randn((geom.shape[0], C), generator=gen, device=dev) * 0.25
    out = plan.launch_forward(feats)                                   # ONE launch over the 8 frames
    assert tuple(out.shape) == (B8, D, H, W, C)
    # fused depth (x) context on the same plan: depth softmax over D bins, context per pixel
    fh, fw = CFG["feature_size"]
    ncam = CFG["num_cameras"]
    dbins = n_frame // (ncam * fh * fw)
    depth = torch.softmax(torch.randn((B8 * ncam, dbins, fh, fw), generator=gen, device=dev), 1)
    ctx = torch.randn((B8 * ncam * fh * fw, C), generator=gen, device=dev) * 0.25
    out_f = plan.launch_fused(depth.reshape(-1), ctx, dbins, fh, fw)
    worst = worst_f = 0.0
    for b in range(B8):
        rows = feats[b * n_frame:(b + 1) * n_frame].cpu().numpy()
        ref = oracle.bev_pool(rows[kept1], coords1[kept1], 1, D, H, W)          # [1, C, D, H, W] float64
        got = out[b].permute(3, 0, 1, 2).cpu().numpy()
        worst = max(worst, float(np.max(np.abs(got - ref[0]))))
        # the explicit outer product of frame b (depth_lss.py:92-97), pooled by the oracle
        d_b = depth[b * ncam:(b + 1) * ncam].reshape(ncam, dbins, fh * fw, 1)
        c_b = ctx[b * ncam * fh * fw:(b + 1) * ncam * fh * fw].reshape(ncam, 1, fh * fw, C)
        vol = (d_b * c_b).reshape(-1, C).cpu().numpy()
        ref_f = oracle.bev_pool(vol[kept1], coords1[kept1], 1, D, H, W)
        worst_f = max(worst_f, float(np.max(np.abs(out_f[b].permute(3, 0, 1, 2).cpu().numpy() - ref_f[0]))))
    assert worst <= 1e-4 and worst_f <= 1e-4, (worst, worst_f)


def test_voxelize_eight_clouds_four_streams_eager_and_graph(dev):
    pts_np = [synth.lidar_points(seed=50 + b) for b in range(B8)]
    pts = [torch.from_numpy(p).to(dev) for p in pts_np]
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
    of, oc, osz = oracle.voxelize_batch(pts_np, vs, pr, mp, mv)

    def check(f, c, sz, tot):
        n = int(tot.item())
        assert n == oc.shape[0]
        assert np.array_equal(c[:n].cpu().numpy(), oc)                 # (batch, x, y, z) of every voxel, packed sample by sample
        assert np.array_equal(sz[:n].cpu().numpy(), osz)
        assert float(np.max(np.abs(f[:n].cpu().numpy() - of))) <= 1e-3

    f, c, sz, tot = voxelize_batch_device(pts, vs, pr, mp, mv)
    check(f, c, sz, tot)
    # the same call captured into a HIP graph (as bench.py replays it) and replayed on fresh inputs
    static = [torch.zeros_like(p) for p in pts]                        # warm-up / capture on all-zero clouds (one voxel each)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        voxelize_batch_device(static, vs, pr, mp, mv)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gf, gc, gsz, gtot = voxelize_batch_device(static, vs, pr, mp, mv)
    for s, p in zip(static, pts):
        s.copy_(p)
    graph.replay()
    torch.cuda.synchronize()
    check(gf, gc, gsz, gtot)
    assert torch.equal(gc[: int(gtot.item())], c[: int(tot.item())])


def test_fused_encoder_eight_frames_vs_module_path_and_oracle_levels(dev):
    pts_np = [synth.lidar_points(seed=70 + b, sweeps=10 if b < 2 else 3) for b in range(B8)]   # two full clouds, six lighter ones
    pts = [torch.from_numpy(p).to(dev) for p in pts_np]
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
    f, c, _, tot = voxelize_batch_device(pts, vs, pr, mp, mv)
    f = f.half()          # what @auto_fp16 hands the module path; the fused path takes either
    enc = flagship_encoder(dev)
    assert fused._variant_for(B8, 27, 64, 64) != 0                   # the batched tilings are the ones under test
    with torch.no_grad():
        got = enc(f, c, B8, num_voxels=tot)
        assert enc.last_path == "fused", enc.last_path_reason
        enc.fused_inference = False
        ref = enc(f, c, B8, num_voxels=tot)
        assert enc.last_path == "modules"
    assert tuple(got.shape) == (B8, 256, 180, 180)
    err = float((got.float() - ref.float()).abs().max())
    record_parity("8 flagship frames: fused encoder vs module path (fp16, dense BEV)", err / (1 + float(ref.float().abs().max())), 4e-4)
    assert err <= 4e-4 * (1 + float(ref.float().abs().max())), err     # observed 1.8e-4 (was 1e-2)
    # level chain of the fused path vs the oracle at 8 frames
    n = int(tot.item())
    ind = c[:n].cpu().numpy()
    lvl = fused.Level(c, c.shape[0], tot.reshape(-1)[:1].int().contiguous(), B8, list(CFG["sparse_shape"]))
    shape = list(CFG["sparse_shape"])
    for ks, st, pd in [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
                       ((3, 3, 3), (2, 2, 2), (1, 1, 0)), ((1, 1, 3), (1, 1, 2), (0, 0, 0))]:
        oi, _, onum, oshape = oracle.get_indice_pairs(ind, B8, shape, ks, st, pd, [1, 1, 1], 0, order="cuda")
        lvl, nbr = lvl.downsample(list(ks), list(st), list(pd))
        m = int(lvl.n_dev.item())
        assert m == oi.shape[0] and np.array_equal(lvl.indices[:m].cpu().numpy(), oi)
        assert np.array_equal((nbr[:, :m] >= 0).sum(1).cpu().numpy(), onum)
        ind, shape = oi, list(oshape)


def test_flagship_frame_stage_by_stage_vs_oracle(dev):
    """Every level of ONE flagship frame: the fused path's own stage input -> (first SubM layer of the level, the strided
    convolution leaving it), raw convolution outputs in fp16 vs oracle.indice_conv in float64 on the same input."""
    from bevfusion_amd.spconv import ops as sops
    from bevfusion_amd.voxel import voxelize_batch

    pts = torch.from_numpy(synth.lidar_points(seed=0)).to(dev)
    feats, coords, _ = voxelize_batch([pts], CFG["voxel_size"], CFG["point_cloud_range"], 10, 160000)
    enc = flagship_encoder(dev)
    rng = np.random.default_rng(0)
    lvl = fused.Level(coords.int().contiguous(), coords.shape[0], None, 1, list(CFG["sparse_shape"]))
    ind, shape = coords.cpu().numpy(), list(CFG["sparse_shape"])
    widths = [16, 32, 64, 128]
    down = [((3, 3, 3), (2, 2, 2), (1, 1, 1), 32), ((3, 3, 3), (2, 2, 2), (1, 1, 1), 64), ((3, 3, 3), (2, 2, 2), (1, 1, 0), 128),
            ((1, 1, 3), (1, 1, 2), (0, 0, 0), 128)]
    x = torch.from_numpy(rng.standard_normal((ind.shape[0], 16)).astype(np.float32) * 0.5).to(dev).half()
    for stage, (cw, (ks, st, pd, cout)) in enumerate(zip(widths, down)):
        n = ind.shape[0]
        # (a) SubM cw -> cw through whatever kernel the fused path picks for this level (slab on the sorted levels)
        w = torch.from_numpy((rng.standard_normal((3, 3, 3, cw, cw)) / np.sqrt(cw * 27 / 4)).astype(np.float32)).to(dev).half()
        conv = type("C", (), dict(subm=True, kernel_size=(3, 3, 3)))()
        variant = fused._slab_variant_for(conv, lvl, cw, cw)
        img = sops.make_filter_image(w)
        if variant is not None:
            got = sops.sparse_conv_slab(x, img, lvl.subm_slab(sops.slab_block_rows(cw, variant)), lvl.n_cap, cw, cw,
                                        num_out_dev=lvl.n_dev, variant=variant)[:n]
        else:
            got = sops.sparse_conv_tiled(x, img, lvl.subm_neighbors((3, 3, 3)), lvl.n_cap, 27, cw, cw, num_out_dev=lvl.n_dev)[:n]
        # level 1 rows are in first-appearance order (gather kernel); round 5: the 128-channel level runs on 64-row slab blocks below
        # 4 frames per step (it kept the gather kernel until round 4)
        assert (variant is not None) == (stage in (1, 2, 3)) and (stage != 3 or variant == 1642220)
        _, sp, sn, _ = oracle.get_indice_pairs(ind, 1, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
        ref = oracle.indice_conv(x[:n].float().cpu().numpy(), w.float().cpu().numpy(), sp, sn, n)
        err = float(np.max(np.abs(got.float().cpu().numpy() - ref)))
        record_parity(f"flagship frame, SubM {cw}->{cw} vs float64 oracle (fp16)", err / (1 + np.abs(ref).max()), 6e-4)
        assert err <= 6e-4 * (1 + np.abs(ref).max()), (stage, "subm", err)       # observed <= 2.5e-4 (was 2e-3)
        # (b) the strided convolution leaving the level
        oi, op, on, oshape = oracle.get_indice_pairs(ind, 1, shape, ks, st, pd, [1, 1, 1], 0, order="cuda")
        nxt, nbr = lvl.downsample(list(ks), list(st), list(pd))
        m = int(nxt.n_dev.item())
        assert m == oi.shape[0] and np.array_equal(nxt.indices[:m].cpu().numpy(), oi)
        K = int(np.prod(ks))
        ws = torch.from_numpy((rng.standard_normal(tuple(ks) + (cw, cout)) / np.sqrt(cw * K / 4)).astype(np.float32)).to(dev).half()
        y = sops.sparse_conv_tiled(x, sops.make_filter_image(ws), nbr, nxt.n_cap, K, cw, cout, num_out_dev=nxt.n_dev,
                                   variant=fused._variant_for(1, K, cw, cout))
        refy = oracle.indice_conv(x[:n].float().cpu().numpy(), ws.float().cpu().numpy(), op, on, m)
        erry = float(np.max(np.abs(y[:m].float().cpu().numpy() - refy)))
        record_parity(f"flagship frame, strided conv leaving level {stage + 1} vs float64 oracle (fp16)", erry / (1 + np.abs(refy).max()), 6e-4)
        assert erry <= 6e-4 * (1 + np.abs(refy).max()), (stage, "strided", erry)   # observed <= 2.6e-4 (was 2e-3)
        # next level: the GPU's own output is the next stage's input (ReLU'd like the network's activations)
        x = torch.relu(y).contiguous()
        x[m:] = 0
        lvl, ind, shape = nxt, oi, list(oshape)
    assert shape == [180, 180, 2]
