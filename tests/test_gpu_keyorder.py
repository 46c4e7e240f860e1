"""Key-ordered level 1 of the LiDAR branch (round 3): the voxelizer writes each sample's voxels in ascending linear cell index
instead of first appearance (same surviving set, same values), level 1 of the SparseEncoder then needs no hash and no int32
neighbour table — sorted-key search builds the slab metadata directly — and its narrow layers (5->16, 16->16, strided 16->32)
run on the staged-rows kernels of csrc/spconv_slab_small.h.

  voxelizer    order="key" == order="first" permuted by linear key (bit-exact: coords, counts, features), cap semantics kept
               (the first max_voxels voxels BY FIRST APPEARANCE survive), ragged / empty samples, vs `oracle.voxelize_batch`;
  index        keys / x-plane directory vs numpy; the status word flags rows that are not ascending;
  metadata     `slab_build_from_sorted` (SubM and strided) == `slab_build` from the int32 neighbour table, byte for byte;
  convolution  narrow slab kernels == gather kernels bit for bit (every built block size, fp16 + bf16, full epilogue,
               blocks whose ranges exceed the staging buffer) and <= 2e-3 * (1 + max) against the float64 oracle;
  encoder      8 flagship frames: dense BEV output of the key-ordered path is BIT-IDENTICAL to the first-appearance path.

References: bevfusion.py:169-197, voxelization_cuda.cu:231-373, spconv_ops.h:27-141 / 260-361, indice.cu.h:147-203."""
import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import synth
from bevfusion_amd.sparse_encoder import SparseEncoder
from bevfusion_amd.spconv import fused
from bevfusion_amd.spconv import ops as sops
from bevfusion_amd.voxel import voxelize_batch, voxelize_batch_device

pytestmark = pytest.mark.gpu
CFG = synth.CL_CONFIG


def linear_key(c4, shape):
    c = c4.astype(np.int64)
    return ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]


def grid_of(cfg=CFG):
    vs, pr = cfg["voxel_size"], cfg["point_cloud_range"]
    return [int(round((pr[3 + i] - pr[i]) / vs[i])) for i in range(3)]


# ---- voxelizer ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_voxels", [160000, 3000])
def test_voxelizer_key_order_is_first_order_permuted(dev, max_voxels):
    pts_np = [synth.lidar_points(seed=3, sweeps=2), np.zeros((0, 5), np.float32), synth.lidar_points(seed=4, sweeps=1)[:777],
              synth.lidar_points(seed=5, sweeps=3)]
    pts = [torch.from_numpy(p).to(dev) for p in pts_np]
    vs, pr, mp = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"]
    g = grid_of()
    f0, c0, s0, t0 = voxelize_batch_device(pts, vs, pr, mp, max_voxels)
    f1, c1, s1, t1 = voxelize_batch_device(pts, vs, pr, mp, max_voxels, order="key")
    n = int(t0.item())
    assert n == int(t1.item()) and n > 0
    c0n, c1n = c0[:n].cpu().numpy(), c1[:n].cpu().numpy()
    k0, k1 = linear_key(c0n, g), linear_key(c1n, g)
    assert np.all(np.diff(k1) > 0)                                    # strictly ascending over the whole packed batch
    perm = np.argsort(k0, kind="stable")
    assert np.array_equal(c0n[perm], c1n)                              # the same surviving set (cap = first appearance)
    assert np.array_equal(s0[:n].cpu().numpy()[perm], s1[:n].cpu().numpy())
    assert np.array_equal(f0[:n].cpu().numpy()[perm], f1[:n].cpu().numpy())   # same sums, same division: bit-exact
    # against the CPU restatement of the reference (first appearance), permuted the same way
    of, oc, osz = oracle.voxelize_batch(pts_np, vs, pr, mp, max_voxels)
    assert oc.shape[0] == n
    op = np.argsort(linear_key(oc, g), kind="stable")
    assert np.array_equal(oc[op], c1n) and np.array_equal(osz[op], s1[:n].cpu().numpy())
    assert float(np.max(np.abs(of[op] - f1[:n].cpu().numpy()))) <= 1e-3
    if max_voxels == 3000:
        assert (np.bincount(c1n[:, 0], minlength=4) == 3000).sum() >= 2   # the cap was hit: survivors are NOT the 3000 smallest keys
    # padded slabs (packed = 0) through voxelize_batch(sync=True)
    fa, ca, sa = voxelize_batch(pts, vs, pr, mp, max_voxels, order="key")
    assert np.array_equal(ca.cpu().numpy(), c1n) and np.array_equal(fa.cpu().numpy(), f1[:n].cpu().numpy())


def test_voxelizer_key_order_replays_from_a_graph(dev):
    pts_np = [synth.lidar_points(seed=60 + b, sweeps=2) for b in range(3)]
    pts = [torch.from_numpy(p).to(dev) for p in pts_np]
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], 40000
    f, c, sz, tot = voxelize_batch_device(pts, vs, pr, mp, mv, order="key")
    static = [torch.zeros_like(p) for p in pts]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        voxelize_batch_device(static, vs, pr, mp, mv, order="key")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gf, gc, gsz, gtot = voxelize_batch_device(static, vs, pr, mp, mv, order="key")
    for s, p in zip(static, pts):
        s.copy_(p)
    graph.replay()
    torch.cuda.synchronize()
    n = int(tot.item())
    assert int(gtot.item()) == n
    assert torch.equal(gc[:n], c[:n]) and torch.equal(gf[:n], f[:n]) and torch.equal(gsz[:n], sz[:n])


# ---- sorted-key index ----------------------------------------------------------------------------------------------------
def random_sorted_set(rng, batch, shape, n, dev):
    vol = batch * shape[0] * shape[1] * shape[2]
    keys = np.sort(rng.choice(vol, size=min(n, vol), replace=False)).astype(np.int64)
    z = keys % shape[2]
    y = (keys // shape[2]) % shape[1]
    x = (keys // (shape[2] * shape[1])) % shape[0]
    b = keys // (shape[2] * shape[1] * shape[0])
    c4 = np.stack([b, x, y, z], 1).astype(np.int32)
    return c4, keys, torch.from_numpy(c4).to(dev)


def split_sorted_index(index, n_cap, batch, shape):
    off = (max(n_cap, 1) * 4 + 255) // 256 * 256
    raw = index.cpu().numpy()
    keys = raw[: n_cap * 4].view(np.uint32)
    xstart = raw[off: off + (batch * shape[0] + 1) * 4].view(np.int32)
    return keys, xstart


@pytest.mark.parametrize("batch,shape,n", [(1, [7, 9, 5], 60), (3, [40, 33, 11], 5000), (2, [16, 16, 4], 2048), (2, [5, 4, 3], 1)])
def test_sorted_index_vs_numpy(dev, batch, shape, n):
    rng = np.random.default_rng(n)
    c4, keys, ct = random_sorted_set(rng, batch, shape, n, dev)
    m = c4.shape[0]
    cap = m + 37                                                        # capacity-padded, live count on the device
    buf = torch.full((cap, 4), -7, dtype=torch.int32, device=dev)
    buf[:m] = ct
    n_dev = torch.tensor([m], dtype=torch.int32, device=dev)
    index, status = sops.sorted_index_build(buf, cap, n_dev, batch, shape)
    gk, gx = split_sorted_index(index, cap, batch, shape)
    assert np.array_equal(gk[:m].astype(np.int64), keys)
    planes = keys // (shape[1] * shape[2])
    ref = np.searchsorted(planes, np.arange(batch * shape[0] + 1), side="left").astype(np.int32)
    assert np.array_equal(gx, ref)
    assert int(status.item()) == 0
    # an empty set: every plane empty
    zero = torch.zeros(1, dtype=torch.int32, device=dev)
    index, status = sops.sorted_index_build(buf, cap, zero, batch, shape)
    assert not split_sorted_index(index, cap, batch, shape)[1].any() and int(status.item()) == 0
    if m > 2:      # broken promise: swap two rows
        bad = buf.clone()
        bad[[0, m - 1]] = bad[[m - 1, 0]]
        _, status = sops.sorted_index_build(bad, cap, n_dev, batch, shape)
        assert int(status.item()) & 2


# ---- metadata ------------------------------------------------------------------------------------------------------------
def meta_bytes(meta, m, rows):
    nblk = (m + rows - 1) // rows
    hdr = meta.hdr.cpu().numpy()[: nblk * 3 * 8].view(np.int32).reshape(nblk, 3, 2)
    slots = meta.slots.cpu().numpy()[: nblk * 27 * rows * 2].view(np.uint16).reshape(nblk, 27, rows)
    return hdr, slots


SETS = [(1, [12, 10, 9], 300), (2, [24, 20, 9], 2500), (1, [6, 10, 80], 4800), (3, [31, 17, 5], 4000), (2, [9, 7, 3], 5)]


@pytest.mark.parametrize("batch,shape,n", SETS)
@pytest.mark.parametrize("rows", [128, 256])
def test_subm_metadata_from_sorted_keys_equals_the_table_route(dev, batch, shape, n, rows):
    rng = np.random.default_rng(7 * n + rows)
    c4, _, ct = random_sorted_set(rng, batch, shape, n, dev)
    m = c4.shape[0]
    cap = m + 300
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    buf[:m] = ct
    n_dev = torch.tensor([m], dtype=torch.int32, device=dev)
    lvl = fused.Level(buf, cap, n_dev, batch, shape)                    # hash index + int32 table: the established route
    ref = sops.slab_build(lvl.subm_neighbors((3, 3, 3)), cap, n_dev, rows)
    index, status = sops.sorted_index_build(buf, cap, n_dev, batch, shape)
    got = sops.slab_build_from_sorted(buf, cap, n_dev, batch, shape, shape, [1, 1, 1], [1, 1, 1], True, index, cap, rows)
    rh, rs = meta_bytes(ref, m, rows)
    gh, gs = meta_bytes(got, m, rows)
    assert np.array_equal(rh, gh) and np.array_equal(rs, gs)
    assert int(got.status.item()) == 0 and int(status.item()) == 0


@pytest.mark.parametrize("batch,shape,n", SETS[:4])
@pytest.mark.parametrize("stride,pad", [((2, 2, 2), (1, 1, 1)), ((2, 2, 2), (1, 1, 0)), ((1, 2, 1), (0, 1, 1))])
@pytest.mark.parametrize("rows", [128, 256])
def test_strided_metadata_from_sorted_keys_equals_the_table_route(dev, batch, shape, n, stride, pad, rows):
    rng = np.random.default_rng(11 * n + rows + stride[0])
    c4, _, ct = random_sorted_set(rng, batch, shape, n, dev)
    m = c4.shape[0]
    buf = torch.zeros((m + 100, 4), dtype=torch.int32, device=dev)
    buf[:m] = ct
    n_dev = torch.tensor([m], dtype=torch.int32, device=dev)
    lvl = fused.Level(buf, buf.shape[0], n_dev, batch, shape, linear_order=True)
    out, nbr = lvl.downsample([3, 3, 3], list(stride), list(pad))      # rank-index route: table built from the input side
    mo = int(out.n_dev.item())
    oi, _, _, _ = oracle.get_indice_pairs(c4, batch, shape, (3, 3, 3), stride, pad, [1, 1, 1], 0, order="cuda")
    assert mo == oi.shape[0] and np.array_equal(out.indices[:mo].cpu().numpy(), oi)
    ref = sops.slab_build(nbr, out.n_cap, out.n_dev, rows)
    got = lvl.down_slab([3, 3, 3], list(stride), list(pad), rows)
    rh, rs = meta_bytes(ref, mo, rows)
    gh, gs = meta_bytes(got, mo, rows)
    assert np.array_equal(rh, gh) and np.array_equal(rs, gs)
    assert fused.geometry_status(lvl) == 0


# ---- convolution ------------------------------------------------------------------------------------------------------------
def _filters(rng, ks, cin, cout, dev, dtype):
    w = rng.standard_normal(tuple(ks) + (cin, cout)) / np.sqrt(cin * 27 / 4)
    return torch.from_numpy(w.astype(np.float32)).to(dev).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16)])
@pytest.mark.parametrize("variant", [3000256, 3000128])
@pytest.mark.parametrize("batch,shape,n", [(2, [24, 20, 9], 2500), (1, [6, 10, 80], 4800), (1, [40, 40, 21], 9000)])
def test_narrow_subm_slab_is_the_gather_kernel_bit_for_bit(dev, dtype, cin, cout, variant, batch, shape, n):
    """(1, [6, 10, 80], dense) makes plane ranges of 256 + 2*81 rows: past the 384-row staging buffer -> the global-memory
    operand path of a block; the other sets stay resident."""
    rng = np.random.default_rng(n + cin)
    c4, _, ct = random_sorted_set(rng, batch, shape, n, dev)
    m = c4.shape[0]
    cap = m + 129
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    buf[:m] = ct
    n_dev = torch.tensor([m], dtype=torch.int32, device=dev)
    lvl = fused.Level(buf, cap, n_dev, batch, shape, linear_order=True)
    pitch = sops.padded_channels(cin)
    x = torch.zeros((cap, pitch), dtype=dtype, device=dev)
    x[:m, :cin] = torch.from_numpy(rng.standard_normal((m, cin)).astype(np.float32) * 0.5).to(dev).to(dtype)
    x[m:] = float("nan")                                                # dead rows must never be read
    w = _filters(rng, (3, 3, 3), cin, cout, dev, dtype)
    img = sops.make_filter_image(w)
    rows = sops.slab_block_rows(cin, variant)
    assert rows == variant - 3000000
    meta = lvl.subm_slab(rows)
    nbr = lvl.subm_neighbors((3, 3, 3))
    bias = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1).to(dev).to(dtype)
    scale = torch.from_numpy(rng.uniform(0.7, 1.3, cout).astype(np.float32)).to(dev)
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1).to(dev)
    res = torch.from_numpy(rng.standard_normal((cap, cout)).astype(np.float32) * 0.3).to(dev).to(dtype)
    for kw in (dict(), dict(bias=bias, bn_scale=scale, bn_shift=shift, residual=res, relu=True)):
        got = sops.sparse_conv_slab(x, img, meta, cap, cin, cout, num_out_dev=n_dev, variant=variant, **kw)[:m]
        ref = sops.sparse_conv_tiled(x, img, nbr, cap, 27, cin, cout, num_out_dev=n_dev, **kw)[:m]
        assert torch.equal(got, ref), (kw.keys(), float((got.float() - ref.float()).abs().max()))
    # and the raw convolution against the float64 oracle
    _, sp, sn, _ = oracle.get_indice_pairs(c4, batch, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
    o = oracle.indice_conv(x[:m, :cin].float().cpu().numpy(), w.float().cpu().numpy(), sp, sn, m)
    got = sops.sparse_conv_slab(x, img, meta, cap, cin, cout, num_out_dev=n_dev, variant=variant)[:m]
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    assert float(np.max(np.abs(got.float().cpu().numpy() - o))) <= tol * (1 + np.abs(o).max())
    assert fused.geometry_status(lvl) == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("variant", [3000256, 3000128, 3100256, 3100128])   # 31xxxxx: both output tiles in one wave
@pytest.mark.parametrize("batch,shape,n,pad", [(2, [24, 20, 9], 2500, (1, 1, 1)), (1, [16, 24, 41], 15000, (1, 1, 1)),
                                                (1, [40, 40, 21], 9000, (1, 1, 0))])
def test_narrow_strided_slab_is_the_gather_kernel_bit_for_bit(dev, dtype, variant, batch, shape, n, pad):
    """16 -> 32, stride 2.  (1, [16, 24, 41], 15000) is ~95 % dense: a block's input ranges exceed the staging buffer."""
    rng = np.random.default_rng(n)
    c4, _, ct = random_sorted_set(rng, batch, shape, n, dev)
    m = c4.shape[0]
    cap = m + 50
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    buf[:m] = ct
    n_dev = torch.tensor([m], dtype=torch.int32, device=dev)
    lvl = fused.Level(buf, cap, n_dev, batch, shape, linear_order=True)
    x = torch.from_numpy(rng.standard_normal((cap, 16)).astype(np.float32) * 0.5).to(dev).to(dtype)
    x[m:] = float("nan")
    w = _filters(rng, (3, 3, 3), 16, 32, dev, dtype)
    img = sops.make_filter_image(w)
    rows = sops.slab_block_rows(16, variant)
    out, nbr = lvl.downsample([3, 3, 3], [2, 2, 2], list(pad))
    meta = lvl.down_slab([3, 3, 3], [2, 2, 2], list(pad), rows)
    mo = int(out.n_dev.item())
    scale = torch.from_numpy(rng.uniform(0.7, 1.3, 32).astype(np.float32)).to(dev)
    shift = torch.from_numpy(rng.standard_normal(32).astype(np.float32) * 0.1).to(dev)
    for kw in (dict(), dict(bn_scale=scale, bn_shift=shift, relu=True)):
        got = sops.sparse_conv_slab(x, img, meta, out.n_cap, 16, 32, num_out_dev=out.n_dev, variant=variant, **kw)[:mo]
        ref = sops.sparse_conv_tiled(x, img, nbr, out.n_cap, 27, 16, 32, num_out_dev=out.n_dev, **kw)[:mo]
        assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    oi, op, on, _ = oracle.get_indice_pairs(c4, batch, shape, (3, 3, 3), (2, 2, 2), pad, [1, 1, 1], 0, order="cuda")
    o = oracle.indice_conv(x[:m].float().cpu().numpy(), w.float().cpu().numpy(), op, on, mo)
    got = sops.sparse_conv_slab(x, img, meta, out.n_cap, 16, 32, num_out_dev=out.n_dev, variant=variant)[:mo]
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    assert float(np.max(np.abs(got.float().cpu().numpy() - o))) <= tol * (1 + np.abs(o).max())
    assert fused.geometry_status(lvl) == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128), (16, 32)])
@pytest.mark.parametrize("batch,shape,n,pad", [(2, [24, 20, 9], 2500, (1, 1, 1)), (1, [40, 40, 21], 9000, (1, 1, 0)),
                                                (1, [16, 24, 41], 15000, (1, 1, 1))])
def test_gather_kernels_reading_slot_metadata_equal_the_table_route(dev, dtype, cin, cout, batch, shape, n, pad):
    """The strided layers that stay on the gather kernels (32 -> 64, 64 -> 128) decode `plane start + 16-bit slot` while they load
    a tile's rulebook instead of reading an int32 table: same kernel, same output bits."""
    rng = np.random.default_rng(n + cin)
    c4, _, ct = random_sorted_set(rng, batch, shape, n, dev)
    m = c4.shape[0]
    cap = m + 77
    buf = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    buf[:m] = ct
    n_dev = torch.tensor([m], dtype=torch.int32, device=dev)
    lvl = fused.Level(buf, cap, n_dev, batch, shape, linear_order=True)
    x = torch.from_numpy(rng.standard_normal((cap, cin)).astype(np.float32) * 0.5).to(dev).to(dtype)
    x[m:] = float("nan")
    w = _filters(rng, (3, 3, 3), cin, cout, dev, dtype)
    img = sops.make_filter_image(w)
    out, nbr = lvl.downsample([3, 3, 3], [2, 2, 2], list(pad))
    meta = lvl.down_slab([3, 3, 3], [2, 2, 2], list(pad), 128)
    mo = int(out.n_dev.item())
    scale = torch.from_numpy(rng.uniform(0.7, 1.3, cout).astype(np.float32)).to(dev)
    shift = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1).to(dev)
    for variant in (0, fused._variant_for(8, 27, cin, cout)):
        for kw in (dict(), dict(bn_scale=scale, bn_shift=shift, relu=True)):
            got = sops.sparse_conv_tiled_slots(x, img, meta, out.n_cap, cin, cout, num_out_dev=out.n_dev, variant=variant, **kw)[:mo]
            ref = sops.sparse_conv_tiled(x, img, nbr, out.n_cap, 27, cin, cout, num_out_dev=out.n_dev, variant=variant, **kw)[:mo]
            assert torch.equal(got, ref), (variant, float((got.float() - ref.float()).abs().max()))
    assert fused.geometry_status(lvl) == 0


# ---- encoder ----------------------------------------------------------------------------------------------------------------
def flagship_encoder(dev, dtype=torch.float16, seed=0):
    torch.manual_seed(seed)
    enc = SparseEncoder(5, list(CFG["sparse_shape"]), order=["conv", "norm", "act"], output_channels=128,
                        encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                        encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock")
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.7, 1.3)
            m.weight.data.uniform_(0.7, 1.3)
            m.bias.data.normal_(0, 0.1)
    return enc.to(dev).to(dtype).eval()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_encoder_key_ordered_path_is_bit_identical_to_first_appearance_path(dev, dtype):
    B = 8
    pts = [torch.from_numpy(synth.lidar_points(seed=70 + b, sweeps=10 if b < 2 else 3)).to(dev) for b in range(B)]
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
    enc = flagship_encoder(dev, dtype)
    f0, c0, _, t0 = voxelize_batch_device(pts, vs, pr, mp, mv)
    f1, c1, _, t1 = voxelize_batch_device(pts, vs, pr, mp, mv, order="key")
    with torch.no_grad():
        ref = enc(f0, c0, B, num_voxels=t0)
        assert enc.last_path == "fused", enc.last_path_reason
        fused.LAYER_PROFILE = []
        try:
            got = enc(f1, c1, B, num_voxels=t1, coors_order="linear")
            kinds = [(r["cin"], r["cout"], r["subm"], r["kernel"], r["variant"]) for r in fused.LAYER_PROFILE]
        finally:
            fused.LAYER_PROFILE = None
        assert enc.last_path == "fused", enc.last_path_reason
        again = enc(f1, c1, B, num_voxels=t1, coors_order="linear")      # the un-profiled pass (no int32 tables at level 1)
        lvl = enc.prepare_geometry(c1, B, num_voxels=t1, coors_order="linear")
        prepared = enc(f1, c1, B, num_voxels=t1, geometry=lvl)
    assert tuple(got.shape) == (B, 256, 180, 180)
    # level 1 ran on the narrow slab kernels: input layer, four 16 -> 16 layers, the strided 16 -> 32
    assert kinds[0][:4] == (5, 16, True, "slab") and all(k[3] == "slab" for k in kinds[1:5]) and kinds[5][:4] == (16, 32, False, "slab")
    assert kinds[5][4] == 3100128 and kinds[1][4] == 3000256
    assert torch.equal(again, ref), "un-profiled key-order pass"
    assert torch.equal(prepared, ref), "key-order pass over a prepared geometry"
    assert torch.equal(got, ref), "profiled key-order pass"
    assert fused.geometry_status(lvl) == 0
    # the promise is checked on the device: first-appearance rows passed off as linear raise the status bit
    lvl_bad = fused.Level(c0, c0.shape[0], t0.reshape(-1)[:1].int().contiguous(), B, list(CFG["sparse_shape"]), linear_order=True)
    lvl_bad.ensure_sorted()
    assert fused.geometry_status(lvl_bad) & 2


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_voxelizer_writes_the_encoder_rows_itself(dev, dtype):
    """`voxelize_batch_device(encoder_rows=dtype)`: the mean kernel also writes the 16-bit, zero-padded rows the first convolution
    reads — bit for bit what the encoder's own pad-and-cast pass makes of the fp32 means — and the encoder takes them as they are:
    same dense output, one launch and one pass over the rows less in the LiDAR branch."""
    B = 3
    pts = [torch.from_numpy(synth.lidar_points(seed=60 + b, sweeps=2 if b else 1)).to(dev) for b in range(B)]
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
    for order in ("key", "first"):
        f, c, _, t = voxelize_batch_device(pts, vs, pr, mp, mv, order=order)
        r16, c2, _, t2 = voxelize_batch_device(pts, vs, pr, mp, mv, order=order, encoder_rows=dtype)
        n = int(t.item())
        assert n == int(t2.item()) and torch.equal(c[:n], c2[:n])
        assert r16.dtype == dtype and tuple(r16.shape) == (B * mv, 8)
        want = torch.zeros((n, 8), dtype=dtype, device=dev)
        want[:, :5] = f[:n].to(dtype)
        assert torch.equal(r16[:n].view(torch.int16), want.view(torch.int16))
        enc = flagship_encoder(dev, dtype)
        with torch.no_grad():
            kw = dict(coors_order="linear") if order == "key" else {}
            ref = enc(f, c, B, num_voxels=t, **kw)
            got = enc(r16, c2, B, num_voxels=t2, **kw)
            assert enc.last_path == "fused", enc.last_path_reason
        assert torch.equal(got, ref)
    with pytest.raises(ValueError, match="5 point features"):
        voxelize_batch_device([p[:, :4].contiguous() for p in pts], vs, pr, mp, mv, encoder_rows=dtype)


def test_broken_linear_promise_is_memory_safe_and_the_encoder_falls_back(dev):
    """first-appearance rows (and rows whose coordinates leave the grid) passed off as coors_order="linear".  The
    sorted-key SEARCH runs on the broken index (clamped reads: no fault), the status bit is raised, and the encoder's first
    eager call notices, warns and re-runs on the order-free route: same output as the honest first-appearance call."""
    B = 2
    pts = [torch.from_numpy(synth.lidar_points(seed=80 + b, sweeps=2)).to(dev) for b in range(B)]
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
    f0, c0, _, t0 = voxelize_batch_device(pts, vs, pr, mp, mv)
    n = int(t0.item())
    shape = list(CFG["sparse_shape"])
    nd = t0.reshape(-1)[:1].int().contiguous()
    for corrupt in (False, True):
        c = c0.clone()
        if corrupt:   # coordinates outside the grid / negative batch index in a few rows
            c[5, 1], c[n // 2, 0], c[n - 1, 1], c[7, 3] = shape[0] + 40000, -3, -7, 1 << 20
        lvl_bad = fused.Level(c, c.shape[0], nd, B, shape, linear_order=True)
        lvl_bad.ensure_sorted()
        for rows in (128, 256):
            lvl_bad.subm_slab(rows)                      # the search itself, on the unsorted directory
        lvl_bad.down_slab([3, 3, 3], [2, 2, 2], [1, 1, 1], 128)
        torch.cuda.synchronize()                         # a fault would surface here
        assert fused.geometry_status(lvl_bad) & 2
    enc = flagship_encoder(dev)
    with torch.no_grad():
        ref = enc(f0, c0, B, num_voxels=t0)
        enc2 = flagship_encoder(dev)
        with pytest.warns(RuntimeWarning, match="not in ascending linear index"):
            got = enc2(f0, c0, B, num_voxels=t0, coors_order="linear")
        with pytest.warns(RuntimeWarning, match="not in ascending linear index"):
            lvl = enc2.prepare_geometry(c0, B, num_voxels=t0, coors_order="linear")
        assert not lvl.linear_order
        prepared = enc2(f0, c0, B, num_voxels=t0, geometry=lvl)
    assert torch.equal(got, ref) and torch.equal(prepared, ref)


def test_tilings_follow_the_live_row_count_not_the_frame_count(dev):
    """Eight SPARSE frames (one sweep each: ~1/7 of the capped flagship frame) are tiled like the ~1.1 capped frames they amount to,
    not like 8: the 32-channel layers take the small-batch kernel, the 128-channel layers take 64-row blocks
    (round 5; the gather kernels before); eight capped frames keep the 8-frame kernels.  The first eager call measures the figure once per batch size."""
    B = 8
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]

    def kinds_of(sweeps):
        pts = [torch.from_numpy(synth.lidar_points(seed=40 + b, sweeps=sweeps)).to(dev) for b in range(B)]
        f, c, _, t = voxelize_batch_device(pts, vs, pr, mp, mv, order="key")
        enc = flagship_encoder(dev)
        with torch.no_grad():
            enc(f, c, B, num_voxels=t, coors_order="linear")          # measures the frames-equivalent of this input
            fused.LAYER_PROFILE = []
            try:
                out = enc(f, c, B, num_voxels=t, coors_order="linear")
                kinds = {(r["cin"], r["cout"], r["subm"]): (r["kernel"], r["variant"]) for r in fused.LAYER_PROFILE}
            finally:
                fused.LAYER_PROFILE = None
            enc.fused_inference = False
            ref = enc(f[: int(t.item())].half(), c[: int(t.item())], B)
        assert float((out.float() - ref.float()).abs().max()) <= 2e-2 * (1 + float(ref.abs().max()))
        return kinds, enc.__dict__["_bevamd_frames_hint"][B]

    sparse, fs = kinds_of(1)
    dense, fd = kinds_of(10)
    assert 0.5 < fs < 1.5 and fd == 8.0        # (the 32-channel switch-over is at 1.5 frames, the 128-channel one at 2.5) one sweep: ~23 k voxels per frame, 8 of them ~ one capped frame
    assert sparse[(32, 32, True)] == ("slab", 1322410) and dense[(32, 32, True)] == ("slab", 4100128)
    assert sparse[(128, 128, True)] == ("slab", 1642220) and dense[(128, 128, True)] == ("slab", 1644220)   # 64-row blocks when rows are few
    assert sparse[(64, 64, True)] == dense[(64, 64, True)] == ("slab", 1644228)


def test_encoder_key_ordered_path_replays_from_a_graph(dev):
    B = 2
    pts = [torch.from_numpy(synth.lidar_points(seed=90 + b, sweeps=2)).to(dev) for b in range(B)]
    vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
    enc = flagship_encoder(dev)

    def branch(p):
        f, c, _, t = voxelize_batch_device(p, vs, pr, mp, mv, order="key")
        with torch.no_grad():
            return enc(f, c, B, num_voxels=t, coors_order="linear")

    ref = branch(pts).clone()
    static = [torch.zeros_like(p) for p in pts]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        branch(static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = branch(static)
    for s, p in zip(static, pts):
        s.copy_(p)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
