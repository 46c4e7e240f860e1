"""GPU parity of the sync-free sparse-encoder inference path (spconv/fused.py): rank-index rulebooks against the
CPU oracle (bit-exact indices / neighbour tables), the dense BEV gather against SparseConvTensor.dense(), and the
whole fused encoder against the module-by-module path that mirrors the reference."""
import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import spconv
from bevfusion_amd.sparse_encoder import SparseEncoder
from bevfusion_amd.spconv import fused
from bevfusion_amd.spconv import ops as sops

pytestmark = pytest.mark.gpu


def _coords(rng, B, shape, n):
    idx = []
    for b in range(B):
        lin = rng.choice(int(np.prod(shape)), size=min(n, int(np.prod(shape))), replace=False)
        idx.append(np.concatenate([np.full((len(lin), 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    ind = np.concatenate(idx).astype(np.int32)
    rng.shuffle(ind, axis=0)
    return ind


def _nbr_from_oracle(opairs, onum, K, m):
    nbr = np.full((K, m), -1, np.int32)
    for k in range(K):
        p = opairs[k][:, : int(onum[k])]
        nbr[k, p[1]] = p[0]
    return nbr


@pytest.mark.parametrize("pad_rows", [0, 777])
def test_level_chain_matches_oracle(dev, pad_rows):
    """Hash level -> strided conv (rank index) -> SubM through the rank index -> strided conv through the rank index:
    active sets, their order, device counts and every neighbour table equal the oracle's."""
    rng = np.random.default_rng(11)
    B, shape = 2, (33, 30, 17)
    ind = _coords(rng, B, shape, 2600)
    n = ind.shape[0]
    cap = n + pad_rows
    buf = np.full((cap, 4), 123456, np.int32)        # garbage beyond the live rows
    buf[:n] = ind
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev) if pad_rows else None
    lvl = fused.Level(torch.from_numpy(buf).to(dev), cap, n_dev, B, shape)
    # SubM at level 1 (hash index)
    _, op, on, _ = oracle.get_indice_pairs(ind, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
    got = lvl.subm_neighbors([3, 3, 3]).cpu().numpy()[:, :n]
    assert np.array_equal(got, _nbr_from_oracle(op, on, 27, n))
    cur_ind, cur_shape, cur = ind, list(shape), lvl
    for ks, st, pd in [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 0)), ((1, 1, 3), (1, 1, 2), (0, 0, 0))]:
        oi, op, on, oshape = oracle.get_indice_pairs(cur_ind, B, cur_shape, ks, st, pd, [1, 1, 1], 0, order="cuda")
        nxt, nbr = cur.downsample(list(ks), list(st), list(pd))
        m = int(nxt.n_dev.item())
        assert m == oi.shape[0] and nxt.shape == list(oshape)
        assert np.array_equal(nxt.indices.cpu().numpy()[:m], oi)
        K = int(np.prod(ks))
        assert np.array_equal(nbr.cpu().numpy()[:, :m], _nbr_from_oracle(op, on, K, m))
        # SubM on the new level: lookups through the rank index
        _, sp, sn, _ = oracle.get_indice_pairs(oi, B, list(oshape), (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
        got = nxt.subm_neighbors([3, 3, 3]).cpu().numpy()[:, :m]
        assert np.array_equal(got, _nbr_from_oracle(sp, sn, 27, m))
        cur_ind, cur_shape, cur = oi, list(oshape), nxt


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_dense_bev_gather_equals_dense_permute(dev, dtype):
    rng = np.random.default_rng(2)
    B, shape, C = 2, (21, 70, 3), 24
    ind = _coords(rng, B, shape, 900)
    f = torch.from_numpy(rng.standard_normal((ind.shape[0], C)).astype(np.float32)).to(dev).to(dtype)
    t = spconv.SparseConvTensor(f, torch.from_numpy(ind).to(dev), list(shape), B)
    d = t.dense()
    N, Cc, H, W, D = d.shape
    want = d.permute(0, 1, 4, 2, 3).contiguous().view(N, Cc * D, H, W)
    # hash index (arbitrary row order)
    lvl = fused.Level(torch.from_numpy(ind).to(dev), ind.shape[0], None, B, shape)
    got = fused.dense_bev(fused.FusedTensor(f, lvl))
    assert torch.equal(got, want)
    # rank index: rows in ascending order as a strided conv produces them (1x1x1 'downsample' keeps the set)
    lvl2, nbr = lvl.downsample([1, 1, 1], [1, 1, 1], [0, 0, 0])
    m = int(lvl2.n_dev.item())
    assert m == ind.shape[0]
    f2 = torch.zeros((lvl2.n_cap, C), dtype=dtype, device=dev)
    f2[:m] = f[nbr[0, :m].long()]
    got2 = fused.dense_bev(fused.FusedTensor(f2, lvl2))
    assert torch.equal(got2, want)


def _small_encoder(dev, dtype, seed=1):
    torch.manual_seed(seed)
    enc = SparseEncoder(5, [40, 40, 41], order=["conv", "norm", "act"], output_channels=32,
                        encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 64], [64, 64]],
                        encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock")
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    return enc.to(dev).to(dtype).eval()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-2), (torch.bfloat16, 6e-2)])
@pytest.mark.parametrize("block_type", ["basicblock", "conv_module"])
def test_fused_encoder_matches_module_path(dev, dtype, tol, block_type):
    rng = np.random.default_rng(5)
    B, shape = 2, (40, 40, 41)
    coors = _coords(rng, B, shape, 2500)
    feats = rng.standard_normal((coors.shape[0], 5)).astype(np.float32)
    if block_type == "basicblock":
        enc = _small_encoder(dev, dtype)
    else:
        torch.manual_seed(3)
        enc = SparseEncoder(5, [40, 40, 41], order=["conv", "norm", "act"]).to(dev).to(dtype).eval()
    x, c = torch.from_numpy(feats).to(dev).to(dtype), torch.from_numpy(coors).to(dev)
    with torch.no_grad():
        assert enc.fused_inference
        got = enc(x, c, B)
        assert enc.fused_inference, "fused path fell back"
        enc.fused_inference = False
        ref = enc(x, c, B)
    assert got.shape == ref.shape and got.dtype == ref.dtype
    err = float((got.float() - ref.float()).abs().max())
    assert err <= tol * (1 + float(ref.float().abs().max())), err
    # same bits run to run
    enc.fused_inference = True
    with torch.no_grad():
        assert torch.equal(enc(x, c, B), got)


def test_fused_encoder_with_capacity_padded_inputs(dev):
    """Inputs as `voxelize_batch(..., sync=False)` hands them over: padded buffers + a device count."""
    rng = np.random.default_rng(8)
    B, shape = 1, (40, 40, 41)
    coors = _coords(rng, B, shape, 3000)
    n, cap = coors.shape[0], 5000
    feats = rng.standard_normal((n, 5)).astype(np.float32)
    enc = _small_encoder(dev, torch.float16)
    x, c = torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev)
    xp = torch.full((cap, 5), float("nan"), device=dev)
    xp[:n] = x
    cp = torch.full((cap, 4), 999999, dtype=torch.int32, device=dev)
    cp[:n] = c
    with torch.no_grad():
        want = enc(x, c, B)
        got = enc(xp, cp, B, num_voxels=torch.tensor([n], dtype=torch.int32, device=dev))
    assert torch.equal(got, want) and torch.isfinite(got).all()


def test_fused_falls_back_for_training_and_fp32(dev):
    enc = _small_encoder(dev, torch.float32)
    x = torch.randn(10, 5, device=dev)
    assert not fused.encoder_supported(enc, x)                      # fp32 -> module path
    enc16 = _small_encoder(dev, torch.float16)
    with torch.no_grad():
        assert fused.encoder_supported(enc16, x)
    assert not fused.encoder_supported(enc16, x)                    # grad enabled
    enc16.train()
    with torch.no_grad():
        assert not fused.encoder_supported(enc16, x)


def test_fused_encoder_is_graph_capturable(dev):
    """No host sync anywhere: the whole encoder records into a HIP graph and replays with new inputs.  Runs in a
    child process under a hard timeout so that a wedged capture cannot stall the suite."""
    import os
    import subprocess
    import sys

    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_probe.py")
    r = subprocess.run([sys.executable, probe], capture_output=True, text=True, timeout=170)
    assert r.returncode == 0 and "GRAPH-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_software_pipeline_across_steps_equals_eager(dev):
    """bench.py --overlap ahead in small (graph_probe.py --ahead): the head (voxelizer + rulebook chain) of batch t + 1 replays on one
    stream while the tail (convolutions) of batch t replays on another, two buffer sets alternating with NEW points every step;
    every step's output equals the eager encoder bit for bit.  Child process under a hard timeout."""
    import os
    import subprocess
    import sys

    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_probe.py")
    r = subprocess.run([sys.executable, probe, "--ahead"], capture_output=True, text=True, timeout=230)
    assert r.returncode == 0 and "AHEAD-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_points_to_bev_batch_of_three_without_host_sync(dev):
    """voxelize_batch_device -> fused encoder (device voxel count) == synchronised voxelize_batch -> encoder, B = 3."""
    from bevfusion_amd.voxel import voxelize_batch, voxelize_batch_device

    vs, pr = [1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 40.0, 40.0, 41.0]
    pts = []
    for b in range(3):
        rng = np.random.default_rng(40 + b)
        pts.append(torch.from_numpy((rng.random((6000 + 700 * b, 5)) * np.array([40, 40, 41, 1, 1])).astype(np.float32)).to(dev))
    enc = _small_encoder(dev, torch.float16)
    with torch.no_grad():
        f0, c0, _ = voxelize_batch(pts, vs, pr, 10, 8000)
        want = enc(f0, c0, 3)
        f1, c1, _, tot = voxelize_batch_device(pts, vs, pr, 10, 8000)
        got = enc(f1, c1, 3, num_voxels=tot)
    assert tuple(got.shape) == (3, 64, 5, 5) and torch.equal(got, want)


def test_empty_frame_goes_through_and_keeps_the_fused_path_enabled(dev):
    enc = _small_encoder(dev, torch.float16)
    with torch.no_grad():
        out = enc(torch.zeros((0, 5), device=dev), torch.zeros((0, 4), dtype=torch.int32, device=dev), 2)
        assert tuple(out.shape) == (2, 64, 5, 5) and float(out.abs().max()) == 0.0
        assert enc.fused_inference
        # a device count of zero with capacity-padded buffers: every level is empty, output is all zeros
        xs = torch.full((100, 5), float("nan"), device=dev)
        cs = torch.full((100, 4), 7, dtype=torch.int32, device=dev)
        out = enc(xs, cs, 2, num_voxels=torch.zeros(1, dtype=torch.int32, device=dev))
        assert tuple(out.shape) == (2, 64, 5, 5) and float(out.abs().max()) == 0.0 and enc.fused_inference


def test_prepared_geometry_path_is_bit_identical(dev):
    """`enc.prepare_geometry(coors, B, num_voxels)` ahead of time + `enc(..., geometry=lvl)` == the one-call fused forward, also with
    the geometry built on ANOTHER stream than the convolutions (how bench.py overlaps it with the camera branch)."""
    rng = np.random.default_rng(9)
    B, shape = 2, (40, 40, 41)
    coors = _coords(rng, B, shape, 2600)
    n, cap = coors.shape[0], 6000
    enc = _small_encoder(dev, torch.float16)
    xp = torch.zeros((cap, 5), device=dev)
    xp[:n] = torch.from_numpy(rng.standard_normal((n, 5)).astype(np.float32)).to(dev)
    cp = torch.full((cap, 4), 77777, dtype=torch.int32, device=dev)
    cp[:n] = torch.from_numpy(coors).to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    with torch.no_grad():
        want = enc(xp, cp, B, num_voxels=cnt)
        lvl = enc.prepare_geometry(cp, B, num_voxels=cnt)
        got = enc(xp, cp, B, num_voxels=cnt, geometry=lvl)
        assert enc.last_path == "fused" and torch.equal(got, want)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            lvl2 = enc.prepare_geometry(cp, B, num_voxels=cnt)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(enc(xp, cp, B, num_voxels=cnt, geometry=lvl2), want)
        with pytest.raises(RuntimeError):
            enc(xp[:100], cp[:100], B, num_voxels=cnt, geometry=lvl)


@pytest.mark.parametrize("B,shape,n,ks,st,pd", [(3, (33, 30, 17), 2600, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                              (2, (40, 37, 5), 3000, (3, 3, 3), (2, 2, 2), (1, 1, 0)),
                                              (2, (21, 19, 11), 900, (1, 1, 3), (1, 1, 2), (0, 0, 0)),
                                              (1, (9, 9, 9), 729, (3, 3, 3), (2, 2, 2), (1, 1, 1)),      # every cell active
                                              (5, (600, 600, 41), 60000, (3, 3, 3), (2, 2, 2), (1, 1, 1))])  # 2048-word tiles
def test_downsample_of_a_sorted_level_equals_the_general_downsample(dev, B, shape, n, ks, st, pd):
    """bevamd_spconv_downsample_sorted (LDS bitmap tiles from contiguous input row ranges; level 1 in key order: rows found through
    the sorted-key directory, src_kind 0; later levels through their rank index, src_kind 1) = bevamd_spconv_downsample (byte map):
    output rows, count and the outputs' rank-index words, bit for bit, with garbage rows beyond the live count."""
    from bevfusion_amd import _capi

    lib = _capi.load()
    rng = np.random.default_rng(B * 1000 + n)
    ind = _coords(rng, B, shape, n)
    lin = ((ind[:, 0].astype(np.int64) * shape[0] + ind[:, 1]) * shape[1] + ind[:, 2]) * shape[2] + ind[:, 3]
    ind = ind[np.argsort(lin)]
    live = ind.shape[0]
    cap = live + 333
    buf = np.full((cap, 4), 7, np.int32)
    buf[:live] = ind
    coors = torch.from_numpy(buf).to(dev)
    n_dev = torch.tensor([live], dtype=torch.int32, device=dev)

    def chain(sorted_route):
        fused._DOWN_SORTED = sorted_route
        lvl = fused.Level(coors, cap, n_dev, B, list(shape), linear_order=True)
        out, _ = lvl.downsample(list(ks), list(st), list(pd), want_nbr=False)
        out2, _ = out.downsample([3, 3, 3], [2, 2, 2], [1, 1, 1], want_nbr=False)   # through the rank index of `out`
        torch.cuda.synchronize()
        res = []
        for o in (out, out2):
            m = int(o.n_dev.item())
            nw = (B * int(np.prod(o.shape)) + 31) // 32
            words = o.index[: nw * 8].view(torch.int32).reshape(nw, 2).cpu().numpy()
            res.append((m, o.indices[:m].cpu().numpy(), words, o.shape))
        return res

    try:
        a, b = chain(True), chain(False)
    finally:
        fused._DOWN_SORTED = True
    for (ma, ia, wa, sa), (mb, ib, wb, sb) in zip(a, b):
        assert ma == mb and sa == sb and ma > 0
        assert np.array_equal(ia, ib)
        assert np.array_equal(wa, wb)
