"""Autograd functions — mirror of `mmdet3d/ops/spconv/functional.py:22-123`.

`indice_conv / indice_subm_conv / indice_inverse_conv` keep the reference's (features, filters, indice_pairs,
indice_pair_num, num_activate_out) signature; `rulebook_conv` is the native entry the modules call (it carries
the output-stationary table instead of pair lists).  Under autocast the inputs are cast to half, exactly like
`custom_fwd(cast_inputs=torch.half)` in the reference (functional.py:24)."""
import torch
from torch.autograd import Function

from . import ops as ops

try:
    from torch.amp import custom_bwd as _cbwd, custom_fwd as _cfwd

    def custom_fwd(fn):
        return _cfwd(fn, device_type="cuda", cast_inputs=torch.half)

    def custom_bwd(fn):
        return _cbwd(fn, device_type="cuda")
except ImportError:  # pragma: no cover
    from torch.cuda.amp import custom_bwd, custom_fwd as _cfwd_old

    def custom_fwd(fn):
        return _cfwd_old(fn, cast_inputs=torch.half)


class RulebookConvFunction(Function):
    @staticmethod
    @custom_fwd
    def forward(ctx, features, filters, rulebook):
        ctx.rulebook = rulebook
        ctx.save_for_backward(features, filters)
        return ops.sparse_conv(features, filters, rulebook.conv_tables()[0], rulebook.num_out)

    @staticmethod
    @custom_bwd
    def backward(ctx, grad_output):
        features, filters = ctx.saved_tensors
        nbr, nbr_t = ctx.rulebook.conv_tables()
        in_grad, f_grad = ops.sparse_conv_backward(features, filters, grad_output, nbr, nbr_t, features.shape[0], rulebook=ctx.rulebook)
        return in_grad, f_grad.to(filters.dtype), None


class _PairsConvFunction(Function):
    """Reference-shaped variant (pair lists); `inverse`/`subm` are static per subclass."""

    inverse = False
    subm = False

    @classmethod
    def _fwd(cls, ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        return ops.indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, cls.inverse, cls.subm)

    @classmethod
    def _bwd(cls, ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(features, filters, grad_output, indice_pairs, indice_pair_num,
                                                        cls.inverse, cls.subm)
        return input_bp, filters_bp, None, None, None


class SparseConvFunction(_PairsConvFunction):
    @staticmethod
    @custom_fwd
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SparseConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out)

    @staticmethod
    @custom_bwd
    def backward(ctx, grad_output):
        return SparseConvFunction._bwd(ctx, grad_output)


class SparseInverseConvFunction(_PairsConvFunction):
    inverse = True

    @staticmethod
    @custom_fwd
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SparseInverseConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out)

    @staticmethod
    @custom_bwd
    def backward(ctx, grad_output):
        return SparseInverseConvFunction._bwd(ctx, grad_output)


class SubMConvFunction(_PairsConvFunction):
    subm = True

    @staticmethod
    @custom_fwd
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        return SubMConvFunction._fwd(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out)

    @staticmethod
    @custom_bwd
    def backward(ctx, grad_output):
        return SubMConvFunction._bwd(ctx, grad_output)


class SparseMaxPoolFunction(Function):
    """functional.py:100-121 on reference-shaped pairs."""

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out):
        out = ops.indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, out = ctx.saved_tensors
        input_bp = ops.indice_maxpool_backward(features, out, grad_output, indice_pairs, indice_pair_num)
        return input_bp, None, None, None


class RulebookMaxPoolFunction(Function):
    """Native entry of the pooling modules: carries the output-stationary table instead of pair lists."""

    @staticmethod
    def forward(ctx, features, rulebook):
        out = ops.sparse_maxpool(features, rulebook.nbr, rulebook.num_out)
        ctx.rulebook = rulebook
        ctx.save_for_backward(features, out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        features, out = ctx.saved_tensors
        return ops.sparse_maxpool_backward(features, out, grad_output, ctx.rulebook.nbr_transposed()), None


indice_conv = SparseConvFunction.apply
indice_inverse_conv = SparseInverseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_maxpool = SparseMaxPoolFunction.apply
rulebook_conv = RulebookConvFunction.apply
rulebook_maxpool = RulebookMaxPoolFunction.apply
