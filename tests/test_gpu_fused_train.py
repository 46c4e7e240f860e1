"""Training-mode SparseEncoder on the inference kernels (spconv/fused_train.py) against the module-by-module path, which the other
GPU tests pin to the oracle and to the reference's own GPU ops (test_gpu_spconv.py, test_gpu_reference_gpu_goldens.py): layer by
layer (forward, input gradient through the mirrored transposed filter, filter gradient) and the whole encoder (dense output, every
parameter gradient, every BatchNorm buffer).  Reference semantics: ops/spconv/functional.py:22-60 (forward / backward of one
convolution), ops/sparse_block.py:88-107, models/backbones/sparse_encoder.py:100-132."""
import numpy as np
import pytest
import torch

from bevfusion_amd import synth
from bevfusion_amd.sparse_encoder import SparseEncoder
from bevfusion_amd.spconv import conv as spconv_conv
from bevfusion_amd.spconv import fused, fused_train, ops
from bevfusion_amd.voxel import voxelize_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def voxels(dev):
    cfg = synth.CL_CONFIG
    pts = [torch.from_numpy(synth.lidar_points(seed=21 + b, sweeps=2)).to(dev) for b in range(2)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][0], order="key")
    return vf, vc.int().contiguous()


def _rel(a, b):
    a, b = a.detach().float(), b.detach().float()
    return float((a - b).abs().max()) / (1e-6 + float(b.abs().max()))


@pytest.mark.parametrize("c", [16, 32, 64, 128])
def test_subm_layer_forward_and_both_gradients_match_the_gather_kernels(dev, voxels, c):
    """One 3x3x3 SubM layer c -> c over a real level-1 voxel set: slab forward, input gradient on the same kernel with the mirrored
    transposed filter over the same metadata, staged-rows filter gradient — against ops.sparse_conv / sparse_conv_backward over the
    int32 tables of ops.build_rulebook."""
    _, coors = voxels
    n = coors.shape[0]
    shape = list(synth.CL_CONFIG["sparse_shape"])
    g = torch.Generator(device=dev).manual_seed(c)
    x = (torch.randn((n, c), generator=g, device=dev) * 0.5).half()
    gy = (torch.randn((n, c), generator=g, device=dev) * 0.5).half()
    conv = spconv_conv.SubMConv3d(c, c, 3, padding=1, bias=False, indice_key="t").to(dev)
    lvl = fused.Level(coors, n, None, 2, shape, linear_order=True)
    lvl.frames_hint = n / 160000.0
    plan = fused_train._Plan(None, fused_train._Lv(lvl, n), torch.float16, [conv])
    L = fused_train._Layer(conv, plan, plan.lv1, plan.lv1)
    assert L.variant is not None and L.wg_code, (L.variant, L.wg_code)      # the slab kernels are what runs
    L.issue()
    plan.layers.append(L)
    fused_train.prepare_images(plan, dev, stem_needs_grad=True)
    xr = x.clone().requires_grad_(True)
    y = fused_train._LevelConv.apply(xr, conv.weight, L)
    y.backward(gy)
    assert fused.geometry_status(lvl) == 0
    rb = ops.build_rulebook(coors, 2, shape, 3, 1, 1, 1, subm=True)
    w16 = conv.weight.detach().half()
    y_ref = ops.sparse_conv(x, w16, rb.nbr, n)
    nbr, nbr_t = rb.conv_tables()
    dx_ref, dw_ref = ops.sparse_conv_backward(x, w16, gy, nbr, nbr_t, n)
    assert _rel(y, y_ref) <= 4e-3
    assert _rel(xr.grad, dx_ref) <= 4e-3
    assert conv.weight.grad.dtype == torch.float32 and _rel(conv.weight.grad, dw_ref) <= 4e-3


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 64), (64, 128)])
def test_strided_layer_forward_and_both_gradients_match_the_module_kernels(dev, voxels, cin, cout):
    """cin -> 2 cin, 3x3x3 stride 2 over level 1: output set (row order included), forward, input gradient over the transposed table,
    filter gradient on the staged-rows kernel (metadata from the layer's table; the wider layers' ranges exceed a stage and take
    the piece-by-piece path) — against build_rulebook + sparse_conv / sparse_conv_backward."""
    _, coors = voxels
    n = coors.shape[0]
    shape = list(synth.CL_CONFIG["sparse_shape"])
    g = torch.Generator(device=dev).manual_seed(5)
    x = (torch.randn((n, cin), generator=g, device=dev) * 0.5).half()
    conv = spconv_conv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False, indice_key="d").to(dev)
    lvl = fused.Level(coors, n, None, 2, shape, linear_order=True)
    lvl.frames_hint = n / 160000.0
    plan = fused_train._Plan(None, fused_train._Lv(lvl, n), torch.float16, [conv])
    L = fused_train._Layer(conv, plan, plan.lv1, None)
    assert L.wg_code, "the staged-rows filter gradient serves this layer"
    L.issue()
    plan.layers.append(L)
    fused_train.prepare_images(plan, dev, stem_needs_grad=True)
    L.lv_out = fused_train._Lv(lvl.downsample(conv.kernel_size, conv.stride, conv.padding, wait=False, want_nbr=True)[0])
    plan.pending.append(L.lv_out)
    xr = x.clone().requires_grad_(True)
    y = fused_train._LevelConv.apply(xr, conv.weight, L)
    rb = ops.build_rulebook(coors, 2, shape, 3, 2, 1, 1, subm=False)
    m = rb.num_out
    assert L.lv_out.n == m and tuple(y.shape) == (m, cout)
    assert torch.equal(L.lv_out.level.indices[:m], rb.out_indices)
    gy = (torch.randn((m, cout), generator=g, device=dev) * 0.5).half()
    y.backward(gy)
    assert fused.geometry_status(lvl) == 0
    w16 = conv.weight.detach().half()
    y_ref = ops.sparse_conv(x, w16, rb.nbr, m)
    nbr, nbr_t = rb.conv_tables()
    dx_ref, dw_ref = ops.sparse_conv_backward(x, w16, gy, nbr, nbr_t, n)
    assert _rel(y, y_ref) <= 4e-3
    assert _rel(xr.grad, dx_ref) <= 4e-3
    assert _rel(conv.weight.grad, dw_ref) <= 4e-3


def _flagship(dev):
    cfg = synth.CL_CONFIG
    torch.manual_seed(0)
    e = SparseEncoder(5, list(cfg["sparse_shape"]), order=["conv", "norm", "act"], output_channels=128,
                      encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                      encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock")
    return e.to(dev).train()


@pytest.mark.parametrize("amp,out_tol,cos", [(torch.float16, 2e-2, 0.999), (torch.bfloat16, 8e-2, 0.99)])
def test_encoder_training_step_fused_vs_modules(dev, voxels, amp, out_tol, cos):
    """The flagship encoder in train() under autocast (fp16: the reference's default hook; bf16), twice from the same weights: the fused
    training path (rows promised in linear order) and the module path.  Dense output, every parameter gradient, every BatchNorm buffer."""
    vf, vc = voxels
    results = []
    for fused_on in (True, False):
        enc = _flagship(dev)
        enc.fused_training = fused_on
        with torch.autocast("cuda", dtype=amp):
            y = enc(vf, vc, 2, coors_order="linear")
        assert enc.last_path == ("fused-train" if fused_on else "modules"), (enc.last_path, enc.last_path_reason)
        (y.float().square().sum() * 1e-3).backward()
        results.append((y.detach().float(), {k: p.grad.clone() for k, p in enc.named_parameters()},
                        {k: b.clone() for k, b in enc.named_buffers()}))
    (y0, g0, b0), (y1, g1, b1) = results
    assert y0.shape == y1.shape and float((y0 - y1).abs().max()) <= out_tol * (1 + float(y1.abs().max()))
    rt = 2e-3 if amp == torch.float16 else 2e-2
    for k in b1:
        if b1[k].dtype.is_floating_point:
            assert torch.allclose(b0[k], b1[k], rtol=rt, atol=rt * 0.1), k
        else:
            assert torch.equal(b0[k], b1[k]), k
    assert set(g0) == set(g1)
    if amp == torch.float16:
        worst = max(float((g0[k] - g1[k]).abs().max()) / (1e-6 + float(g1[k].abs().max())) for k in g1)
        assert worst <= 0.2, worst
    a = torch.cat([g0[k].reshape(-1).double() for k in g1])
    b = torch.cat([g1[k].reshape(-1).double() for k in g1])
    assert float((a @ b) / (a.norm() * b.norm())) >= cos


def test_fallbacks_leave_the_batchnorm_buffers_alone(dev, voxels):
    """What the fused training path does not serve goes to the module path BEFORE anything ran: rows without the order promise, fp32
    training, rows in first-appearance order passed off as linear (status word, first call)."""
    vf, vc = voxels
    enc = _flagship(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        enc(vf, vc, 2)
    assert enc.last_path == "modules" and "linear order" in enc.last_path_reason
    enc(vf, vc, 2, coors_order="linear")                       # fp32, no autocast
    assert enc.last_path == "modules" and "fp32" in enc.last_path_reason
    enc = _flagship(dev)
    perm = torch.randperm(vc.shape[0], device=dev)
    with torch.autocast("cuda", dtype=torch.float16):
        enc(vf[perm], vc[perm], 2, coors_order="linear")       # a broken promise
    assert enc.last_path == "modules" and "status" in enc.last_path_reason
    nb = {k: int(b) for k, b in enc.named_buffers() if k.endswith("num_batches_tracked")}
    assert set(nb.values()) == {1}                              # every BatchNorm saw exactly one batch


def test_a_step_leaves_no_cyclic_garbage_behind(dev, voxels):
    """A training step's activations and rulebooks are released by reference counting alone (no plan <-> layer cycle): with Python's
    cyclic collector paused — bench.py's timed region, or simply a long gap between collections — device memory returns to where it
    was once the step's outputs and gradients are dropped."""
    import gc

    vf, vc = voxels
    enc = _flagship(dev)

    def step():
        with torch.autocast("cuda", dtype=torch.float16):
            y = enc(vf, vc, 2, coors_order="linear")
        assert enc.last_path == "fused-train"
        (y.float().square().sum() * 1e-3).backward()
        enc.zero_grad(set_to_none=True)

    step()                      # caches (BatchNorm workspaces, status pool, tree check) are filled
    gc.collect()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    gc.disable()
    try:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - base
    finally:
        gc.enable()
    assert grown <= 8 << 20, grown


def test_stem_layer_filter_gradient_on_the_staged_rows_kernel(dev, voxels):
    """The stem (5 -> 16 SubM, rows zero-padded to 8 channels for the narrow forward kernel): forward on the slab kernel, filter gradient
    on the staged-rows kernel with the rows padded to 16 channels (the extra rows of dW dropped) — against the gather kernels; no
    int32 table and no hash index of level 1 is built."""
    _, coors = voxels
    n = coors.shape[0]
    shape = list(synth.CL_CONFIG["sparse_shape"])
    g = torch.Generator(device=dev).manual_seed(55)
    x5 = (torch.randn((n, 5), generator=g, device=dev) * 0.5).half()
    gy = (torch.randn((n, 16), generator=g, device=dev) * 0.5).half()
    conv = spconv_conv.SubMConv3d(5, 16, 3, padding=1, bias=False, indice_key="s").to(dev)
    lvl = fused.Level(coors, n, None, 2, shape, linear_order=True)
    lvl.frames_hint = n / 160000.0
    plan = fused_train._Plan(None, fused_train._Lv(lvl, n), torch.float16, [conv])
    L = fused_train._Layer(conv, plan, plan.lv1, plan.lv1)
    assert L.variant is not None and L.wg_code and L.wg_cin == 16
    L.issue()
    L.issue(forward=False)
    plan.layers.append(L)
    fused_train.prepare_images(plan, dev, stem_needs_grad=False)
    x = torch.nn.functional.pad(x5, (0, 3))
    y = fused_train._LevelConv.apply(x, conv.weight, L)
    y.backward(gy)
    assert lvl.index is None and not lvl._subm                  # neither the hash index nor the int32 table exists
    # a caller that asks for the gradient of the voxel features (nobody in the reference does) gets it through the layer's table
    fused_train.prepare_images(plan, dev, stem_needs_grad=True)
    x5g = x5.clone().requires_grad_(True)
    y2 = fused_train._LevelConv.apply(torch.nn.functional.pad(x5g, (0, 3)), conv.weight, L)
    y2.backward(gy)
    rbx = ops.build_rulebook(coors, 2, shape, 3, 1, 1, 1, subm=True)
    nbx, nbx_t = rbx.conv_tables()
    dx_ref, _ = ops.sparse_conv_backward(x5, conv.weight.detach().half(), gy, nbx, nbx_t, n)
    assert tuple(x5g.grad.shape) == (n, 5) and _rel(x5g.grad, dx_ref) <= 4e-3
    conv.weight.grad = None
    y = fused_train._LevelConv.apply(x, conv.weight, L)
    y.backward(gy)
    rb = ops.build_rulebook(coors, 2, shape, 3, 1, 1, 1, subm=True)
    w16 = conv.weight.detach().half()
    y_ref = ops.sparse_conv(x5, w16, rb.nbr, n)
    nbr, nbr_t = rb.conv_tables()
    _, dw_ref = ops.sparse_conv_backward(x5, w16, gy, nbr, nbr_t, n)
    assert _rel(y, y_ref) <= 4e-3
    assert tuple(conv.weight.grad.shape) == (3, 3, 3, 5, 16) and _rel(conv.weight.grad, dw_ref) <= 4e-3
