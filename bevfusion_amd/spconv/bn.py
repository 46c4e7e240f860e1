"""Training-mode BatchNorm1d (+ ReLU, + residual add) over sparse feature rows on the HIP kernels of csrc/sparse_bn.hip.

The sparse blocks of the reference apply `nn.BatchNorm1d`, `nn.ReLU` and the residual add to the `[N, C]` feature matrix as
separate torch modules (ops/sparse_block.py:88-107, models/backbones/sparse_encoder.py:39: BN1d, eps 1e-3, momentum 0.01).  In
training torch runs four BatchNorm kernels per layer plus the elementwise ones: ~6 of 31.8 ms of the --amp step.
`bn_act` runs the same arithmetic — batch statistics in fp32, every stored tensor of the unfused pipeline rounded once — in two
launches forward and two backward; module tree, parameters, buffers and state-dict names are untouched (the containers call this
for a plain `nn.BatchNorm1d` in train() on GPU features; anything else takes the torch modules)."""
import os

import torch
from torch import nn

from .. import _capi

_NATIVE = os.environ.get("BEVAMD_NATIVE_BN", "1") != "0"
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_WS = {}


def usable(bn, feats, residual=None):
    """True when `bn(feats)` (training mode) can run on the native kernels with identical semantics."""
    if not (_NATIVE and type(bn) is nn.BatchNorm1d and bn.training and feats.is_cuda and feats.dim() == 2):
        return False
    if bn.momentum is None or not bn.track_running_stats or feats.dtype not in _DT or feats.shape[0] < 2:
        return False
    c = feats.shape[1]
    vec = 4 if feats.dtype == torch.float32 else 8
    if c % vec or c > 256 or 256 % (c // vec):
        return False
    if residual is not None and (residual.dtype != feats.dtype or residual.shape != feats.shape):
        return False
    return all(p is None or p.dtype == torch.float32 for p in (bn.weight, bn.bias, bn.running_mean, bn.running_var))


def _workspace(dev, c):
    # keyed by the issuing stream as well: the partial sums and the ticket word of a launch must not be shared by two
    # BatchNorm calls running concurrently on different streams of one device (two models, an eager call beside a graph replay)
    key = (dev.index, int(c), int(torch.cuda.current_stream(dev).cuda_stream))
    if key not in _WS:
        while len(_WS) >= 64:            # bounded: the oldest (device, width, stream) entry goes
            _WS.pop(next(iter(_WS)))
        nbytes = int(_capi.load().bevamd_sparse_bn_workspace_bytes(c))
        _WS[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)   # zeroed once: the ticket word; every launch leaves it zero
    return _WS[key]


def _rows(t):
    return t if t.stride(1) == 1 else t.contiguous()


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, eps, momentum, relu):
        lib = _capi.load()
        x = _rows(x)
        n, c = x.shape
        dev = x.device
        dt = _DT[x.dtype]
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        y = torch.empty((n, c), dtype=x.dtype, device=dev)
        res = _rows(residual) if residual is not None else None
        ws = _workspace(dev, c)
        with torch.cuda.device(dev):
            st = _capi.stream_ptr(dev)
            rc = lib.bevamd_sparse_bn_stats(_capi.ptr(x), dt, n, c, x.stride(0), float(eps), float(momentum), _capi.ptr(mean),
                                            _capi.ptr(invstd), _capi.ptr(running_mean), _capi.ptr(running_var), _capi.ptr(ws),
                                            ws.numel(), st)
            _capi.check(rc, "sparse_bn_stats")
            rc = lib.bevamd_sparse_bn_apply(_capi.ptr(x), dt, n, c, x.stride(0), _capi.ptr(mean), _capi.ptr(invstd),
                                            _capi.ptr(weight), _capi.ptr(bias), _capi.ptr(res), res.stride(0) if res is not None else 0,
                                            int(relu), _capi.ptr(y), y.stride(0), st)
            _capi.check(rc, "sparse_bn_apply")
        ctx.save_for_backward(x, y if relu else None, mean, invstd, weight)
        ctx.relu, ctx.has_res, ctx.has_bias = bool(relu), residual is not None, bias is not None
        ctx.mark_non_differentiable(mean, invstd)
        return y, mean, invstd

    @staticmethod
    def backward(ctx, dy, _dmean, _dinvstd):
        lib = _capi.load()
        x, y, mean, invstd, weight = ctx.saved_tensors
        n, c = x.shape
        dev = x.device
        dt = _DT[x.dtype]
        dy = _rows(dy if dy.dtype == x.dtype else dy.to(x.dtype))
        dx = torch.empty_like(x)
        sum_dz = torch.empty(c, dtype=torch.float32, device=dev)
        sum_dz_xhat = torch.empty(c, dtype=torch.float32, device=dev)
        # d residual = dz: the masked incoming gradient (a new tensor only when the ReLU masks; otherwise dy itself)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.relu) else None
        ws = _workspace(dev, c)
        with torch.cuda.device(dev):
            rc = lib.bevamd_sparse_bn_backward(_capi.ptr(dy), dy.stride(0), _capi.ptr(y), y.stride(0) if y is not None else 0,
                                               _capi.ptr(x), x.stride(0), dt, n, c, int(ctx.relu), _capi.ptr(mean), _capi.ptr(invstd),
                                               _capi.ptr(weight), _capi.ptr(sum_dz), _capi.ptr(sum_dz_xhat), _capi.ptr(dx),
                                               dx.stride(0), _capi.ptr(dres), dres.stride(0) if dres is not None else 0,
                                               _capi.ptr(ws), ws.numel(), _capi.stream_ptr(dev))
        _capi.check(rc, "sparse_bn_backward")
        if ctx.has_res and dres is None:
            dres = dy
        return (dx, sum_dz_xhat if weight is not None else None, sum_dz if ctx.has_bias else None, dres if ctx.has_res else None,
                None, None, None, None, None)


def bn_act(feats, bn, relu=False, residual=None):
    """`relu(bn(feats) + residual)` (each optional) for a `nn.BatchNorm1d` in training mode, on the native kernels.  Updates the
    module's running statistics and `num_batches_tracked` exactly as `bn(feats)` would."""
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    y, _, _ = _BnAct.apply(feats, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)
    return y
