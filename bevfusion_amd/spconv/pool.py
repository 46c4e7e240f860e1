"""Sparse max pooling modules — the interface of `mmdet3d/ops/spconv/pool.py:21-85`.

`SparseMaxPool2d/3d(kernel_size, stride=1, padding=0, dilation=1)` turn a SparseConvTensor into the pooled one: the
rulebook of the equivalent strided convolution picks the active outputs, and each output takes the maximum of the inputs
it covers, floored at zero like the reference (its output buffer starts from zeros, pool_ops.h:33).  One launch for the
pooling, one for its backward (input-stationary, no atomics)."""
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


def _per_axis(v, ndim):
    return list(v) if isinstance(v, (list, tuple)) else [v] * ndim


class SparseMaxPool(SparseModule):
    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        self.ndim = ndim
        self.kernel_size = _per_axis(kernel_size, ndim)
        self.stride = _per_axis(stride, ndim)
        self.padding = _per_axis(padding, ndim)
        self.dilation = _per_axis(dilation, ndim)
        self.subm = subm

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        rb = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size, self.stride,
                                self.padding, self.dilation, self.subm)
        pooled = Fsp.rulebook_maxpool(input.features, rb)
        out = SparseConvTensor(pooled, rb.out_indices, rb.out_spatial_shape, input.batch_size)
        out.indice_dict = input.indice_dict
        out.grid = input.grid
        return out


class SparseMaxPool2d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(2, kernel_size, stride, padding, dilation)


class SparseMaxPool3d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(3, kernel_size, stride, padding, dilation)
