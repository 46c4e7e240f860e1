"""Training-mode BatchNorm1d (+ ReLU, + residual) over sparse feature rows on the HIP kernels (csrc/sparse_bn.hip) against
torch.nn.BatchNorm1d / nn.ReLU / add — the modules the reference applies to SparseConvTensor.features (ops/sparse_block.py:88-107,
models/backbones/sparse_encoder.py:39: eps 1e-3, momentum 0.01).  Bars: fp32 <= 1e-5 relative on outputs, input / parameter
gradients and running statistics; 16-bit rows within one rounding of the storage type; bit-reproducible run to run."""
import numpy as np
import pytest
import torch
from torch import nn

from bevfusion_amd import synth
from bevfusion_amd.sparse_encoder import SparseEncoder
from bevfusion_amd.spconv import bn as native_bn
from bevfusion_amd.voxel import voxelize_batch

pytestmark = pytest.mark.gpu


def reference(x, res, bn, relu):
    y = bn(x)
    if res is not None:
        y = y + res
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,c", [(5, 16), (1000, 32), (70001, 64), (4099, 128), (333, 8)])
@pytest.mark.parametrize("relu,with_res", [(False, False), (True, False), (True, True), (False, True)])
def test_against_torch_modules(dev, dtype, n, c, relu, with_res):
    g = torch.Generator(device=dev).manual_seed(n + c)
    x = (torch.randn((n, c), generator=g, device=dev) * 1.7 + 0.3).to(dtype)
    res = torch.randn((n, c), generator=g, device=dev).to(dtype) if with_res else None
    dy = torch.randn((n, c), generator=g, device=dev).to(dtype)
    mods = []
    for _ in range(2):
        m = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
        with torch.no_grad():
            m.weight.copy_(torch.linspace(0.5, 1.5, c))
            m.bias.copy_(torch.linspace(-0.2, 0.2, c))
            m.running_mean.fill_(0.1)
            m.running_var.fill_(0.9)
        mods.append(m)
    ref_bn, our_bn = mods
    if not native_bn.usable(our_bn, x, res):
        pytest.skip("shape not covered by the native kernels (torch modules run)")
    xr, xo = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    ro = res.clone().requires_grad_(True) if with_res else None
    with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
        yr = reference(xr, rr, ref_bn, relu)
    yo = native_bn.bn_act(xo, our_bn, relu=relu, residual=ro)
    assert yo.dtype == x.dtype and yr.dtype == x.dtype
    yr.backward(dy)
    yo.backward(dy)
    tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]

    def close(a, b, scale=1.0):
        a, b = a.float(), b.float()
        a, b = a.detach(), b.detach()
        return float((a - b).abs().max()) <= tol * scale * (1.0 + float(b.abs().max()))

    assert close(yo, yr)
    assert close(xo.grad, xr.grad)
    if with_res:
        assert close(ro.grad, rr.grad)
    assert close(our_bn.weight.grad, ref_bn.weight.grad, 4.0) and close(our_bn.bias.grad, ref_bn.bias.grad, 4.0)
    assert torch.allclose(our_bn.running_mean, ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(our_bn.running_var, ref_bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(our_bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1


def test_bit_reproducible_and_strided_rows(dev):
    """Same input, same bits (no atomics in the reductions); rows wider than the channel count (a column slice) are accepted."""
    g = torch.Generator(device=dev).manual_seed(1)
    wide = torch.randn((50000, 96), generator=g, device=dev).half()
    x = wide[:, :64]
    assert x.stride(0) == 96
    outs = []
    for _ in range(3):
        m = nn.BatchNorm1d(64, eps=1e-3, momentum=0.01).to(dev).train()
        xi = x.detach().requires_grad_(True)
        y = native_bn.bn_act(xi, m, relu=True)
        y.backward(torch.ones_like(y))
        outs.append((y.detach().clone(), None, m.running_var.clone(), m.weight.grad.clone(), xi.grad.clone()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[2], outs[0][2]) and torch.equal(o[3], outs[0][3])
        assert torch.equal(o[4], outs[0][4])
    ref = nn.BatchNorm1d(64, eps=1e-3, momentum=0.01).to(dev).train()
    yr = torch.relu(ref(x.contiguous()))
    assert float((outs[0][0].float() - yr.float()).abs().max()) <= 2e-3 * (1 + float(yr.abs().max()))


def test_eval_mode_and_host_tensors_take_the_torch_modules(dev):
    m = nn.BatchNorm1d(16).to(dev)
    x = torch.randn(100, 16, device=dev)
    assert native_bn.usable(m, x)
    assert not native_bn.usable(m.eval(), x)                       # eval: running statistics, torch's kernel (or the fused inference path)
    assert not native_bn.usable(nn.BatchNorm1d(16).train(), torch.randn(100, 16))            # host tensors
    assert not native_bn.usable(nn.BatchNorm1d(16, momentum=None).to(dev), x)                # cumulative average: torch
    assert not native_bn.usable(nn.BatchNorm1d(16, track_running_stats=False).to(dev), x)


def test_sparse_encoder_training_step_native_vs_torch_batchnorm(dev):
    """The whole encoder in train() under autocast: native BatchNorm path against the torch-module path (same weights): dense
    output, every parameter gradient and every running statistic agree to 16-bit rounding."""
    cfg = synth.CL_CONFIG
    pts = [torch.from_numpy(synth.lidar_points(seed=11 + b, sweeps=2)).to(dev) for b in range(2)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][0])

    def make():
        torch.manual_seed(0)
        e = SparseEncoder(5, list(cfg["sparse_shape"]), order=["conv", "norm", "act"], output_channels=128,
                          encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                          encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock")
        return e.to(dev).train()

    results = []
    for native in (True, False):
        old, native_bn._NATIVE = native_bn._NATIVE, native
        try:
            enc = make()
            with torch.autocast("cuda", dtype=torch.float16):
                y = enc(vf, vc, 2)
            (y.float().square().sum() * 1e-3).backward()      # a loss scale that keeps the fp16 gradients out of the subnormals
            results.append((y.detach().float(), {k: p.grad.clone() for k, p in enc.named_parameters()},
                            {k: b.clone() for k, b in enc.named_buffers()}))
        finally:
            native_bn._NATIVE = old
    (y0, g0, b0), (y1, g1, b1) = results
    assert float((y0 - y1).abs().max()) <= 2e-2 * (1 + float(y1.abs().max()))
    for k in b1:
        if b1[k].dtype.is_floating_point:
            assert torch.allclose(b0[k], b1[k], rtol=2e-3, atol=2e-4), k
        else:
            assert torch.equal(b0[k], b1[k]), k
    # two fp16 pipelines that round at the same points but sum statistics in a different order: parameter by parameter within
    # a few per cent of the largest entry, and the whole gradient pointing the same way
    worst = max(float((g0[k] - g1[k]).abs().max()) / (1e-6 + float(g1[k].abs().max())) for k in g1)
    assert worst <= 0.2, worst
    a = torch.cat([g0[k].reshape(-1).double() for k in g1])
    b = torch.cat([g1[k].reshape(-1).double() for k in g1])
    assert float((a @ b) / (a.norm() * b.norm())) >= 0.999


def test_statistics_of_channels_with_a_large_mean(dev):
    """|mean| >> std.  E[x^2] - mean^2 on raw fp32 sums cancels (relative error ~1e-7 * mean^2 / var: at mean 300,
    std 0.05 that is the whole variance); the kernels accumulate sums shifted by the tensor's first row, like a Welford pass starts."""
    n, c = 50000, 32
    g = torch.Generator(device=dev).manual_seed(3)
    mean = torch.linspace(-300.0, 300.0, c, device=dev)
    std = torch.linspace(0.05, 2.0, c, device=dev)
    x = torch.randn((n, c), generator=g, device=dev) * std + mean
    mods = [nn.BatchNorm1d(c, eps=1e-5, momentum=0.1).to(dev).train() for _ in range(2)]
    ref_bn, our_bn = mods
    assert native_bn.usable(our_bn, x, None)
    yr = ref_bn(x)
    yo = native_bn.bn_act(x.clone(), our_bn, relu=False, residual=None)
    var64 = x.double().var(0, unbiased=True)
    # running_var = 0.9 * 1 + 0.1 * unbiased variance: pin both implementations to float64
    want = 0.9 + 0.1 * var64
    assert float(((our_bn.running_var.double() - want).abs() / want).max()) <= 1e-5
    assert float(((our_bn.running_var - ref_bn.running_var).abs() / ref_bn.running_var).max()) <= 1e-5
    assert torch.allclose(our_bn.running_mean, ref_bn.running_mean, rtol=1e-6, atol=1e-5)
    # normalised outputs: variance var / (var + eps) per channel, and equal to torch's within fp32 rounding of (x - mean) * invstd at |x| ~ 300
    vb = x.double().var(0, unbiased=False)
    assert float((yo - yr).abs().max()) <= 2e-3 and float((yo.double().var(0, unbiased=False) - vb / (vb + 1e-5)).abs().max()) <= 1e-3
