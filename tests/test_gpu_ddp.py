"""GPU, one rank, backend "nccl" (= RCCL on ROCm): the training gradient path of the hot path under DistributedDataParallel —
the reference's only collective (apis/train.py:48-53 MMDistributedDataParallel; fusion_models/base.py:44-46 loss all-reduce).
The sparse ops have no CPU path, so gloo cannot carry this one; world size 1 still drives DDP's reducer, bucket all-reduce and
hooks through RCCL.  Checked: the all-reduced filter gradient of a sparse convolution against oracle.indice_conv_backward, and
the whole SparseEncoder's gradients under DDP against the unwrapped module (bit-identical: every kernel is deterministic)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import oracle
from bevfusion_amd import spconv
from bevfusion_amd.sparse_encoder import SparseEncoder

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_world(dev):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield
    dist.destroy_process_group()
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)


def _indices(rng, B, shape, n):
    idx = []
    for b in range(B):
        lin = rng.choice(int(np.prod(shape)), size=n, replace=False)
        idx.append(np.concatenate([np.full((n, 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    ind = np.concatenate(idx).astype(np.int32)
    rng.shuffle(ind, axis=0)
    return ind


class _ConvOnFeatures(torch.nn.Module):
    """A sparse convolution as a plain tensor -> tensor module (DDP needs tensor outputs to hook the backward)."""

    def __init__(self, conv, indices, shape, batch):
        super().__init__()
        self.conv, self.indices, self.shape, self.batch = conv, indices, shape, batch

    def forward(self, feats):
        return self.conv(spconv.SparseConvTensor(feats, self.indices, self.shape, self.batch)).features


def test_ddp_allreduced_filter_gradient_vs_oracle(dev, rccl_world):
    rng = np.random.default_rng(21)
    B, shape, cin, cout = 2, (16, 14, 7), 16, 32
    ind = _indices(rng, B, shape, 500)
    ks, st, pd = (3, 3, 3), (2, 2, 2), (1, 1, 1)
    oi, opairs, onum, _ = oracle.get_indice_pairs(ind, B, shape, ks, st, pd, [1, 1, 1], 0, order="cuda")
    w = (rng.standard_normal(ks + (cin, cout)) * 0.1).astype(np.float32)
    f = rng.standard_normal((ind.shape[0], cin)).astype(np.float32)
    og = rng.standard_normal((oi.shape[0], cout)).astype(np.float32)
    gi_ref, gw_ref = oracle.indice_conv_backward(f, w, og, opairs, onum)
    conv = spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
    ddp = torch.nn.parallel.DistributedDataParallel(_ConvOnFeatures(conv, torch.from_numpy(ind).to(dev), list(shape), B),
                                                    device_ids=[dev.index])
    x = torch.from_numpy(f).to(dev).requires_grad_(True)
    out = ddp(x)
    out.backward(torch.from_numpy(og).to(dev))
    torch.cuda.synchronize()
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    assert np.max(np.abs(conv.weight.grad.cpu().numpy() - gw_ref)) <= 2e-4 * (1 + np.abs(gw_ref).max())
    assert np.max(np.abs(x.grad.cpu().numpy() - gi_ref)) <= 5e-5 * (1 + np.abs(gi_ref).max())


def test_sparse_encoder_under_ddp_matches_unwrapped(dev, rccl_world):
    rng = np.random.default_rng(4)
    B, shape = 2, (40, 40, 41)
    coors = torch.from_numpy(_indices(rng, B, shape, 2000)).to(dev)
    feats = torch.from_numpy(rng.standard_normal((coors.shape[0], 5)).astype(np.float32)).to(dev)

    def make():
        torch.manual_seed(1)
        return SparseEncoder(5, list(shape), order=["conv", "norm", "act"], output_channels=32,
                             encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 64], [64, 64]],
                             encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock").to(dev).train()

    ref, enc = make(), make()
    ref(feats, coors, B).square().mean().backward()
    ddp = torch.nn.parallel.DistributedDataParallel(enc, device_ids=[dev.index])
    ddp(feats, coors, B).square().mean().backward()
    torch.cuda.synchronize()
    n = 0
    for (name, p), (_, q) in zip(ref.named_parameters(), enc.named_parameters()):
        assert p.grad is not None and q.grad is not None and torch.isfinite(q.grad).all(), name
        assert torch.equal(p.grad, q.grad), name          # deterministic kernels + a one-rank average
        n += 1
    assert n >= 60 and enc.last_path == "modules"


def test_fused_training_path_under_ddp_matches_unwrapped(dev, rccl_world):
    """The fused training path (spconv/fused_train.py: autocast, rows promised in linear order) under DistributedDataParallel over
    RCCL: one convolution node and one BatchNorm node per layer, so DDP's hooks fire per parameter as the backward proceeds; the
    gradients equal the unwrapped encoder's bit for bit (deterministic kernels + a one-rank average)."""
    from bevfusion_amd import synth
    from bevfusion_amd.voxel import voxelize_batch

    cfg = synth.CL_CONFIG
    pts = [torch.from_numpy(synth.lidar_points(seed=31 + b, sweeps=2)).to(dev) for b in range(2)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][0], order="key")

    def make():
        torch.manual_seed(1)
        return SparseEncoder(5, list(cfg["sparse_shape"]), order=["conv", "norm", "act"], output_channels=128,
                             encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                             encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock").to(dev).train()

    ref, enc = make(), make()
    with torch.autocast("cuda", dtype=torch.float16):
        y = ref(vf, vc, 2, coors_order="linear")
    (y.float().square().sum() * 1e-3).backward()
    assert ref.last_path == "fused-train", ref.last_path_reason
    ddp = torch.nn.parallel.DistributedDataParallel(enc, device_ids=[dev.index])
    with torch.autocast("cuda", dtype=torch.float16):
        y2 = ddp(vf, vc, 2, coors_order="linear")
    (y2.float().square().sum() * 1e-3).backward()
    torch.cuda.synchronize()
    assert enc.last_path == "fused-train", enc.last_path_reason
    assert torch.equal(y, y2)
    n = 0
    for (name, p), (_, q) in zip(ref.named_parameters(), enc.named_parameters()):
        assert p.grad is not None and q.grad is not None and torch.isfinite(q.grad).all(), name
        assert torch.equal(p.grad, q.grad), name
        n += 1
    assert n >= 60
