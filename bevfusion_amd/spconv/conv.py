"""Sparse convolution modules — mirror of `mmdet3d/ops/spconv/conv.py:49-279,425-451`.

Same constructor arguments, attributes, parameter names/shapes (`weight` [kx,ky,kz,Cin,Cout], `bias`) and
registration under their class names in CONV_LAYERS, so `build_conv_layer(dict(type="SubMConv3d",
indice_key=...), cin, cout, k, stride=, padding=, bias=False)` (sparse_block.py:156-168) works and reference
checkpoints load by key.  Differences, all internal:
  * the rulebook is the output-stationary `ops.Rulebook` built by hash instead of a dense grid;
  * rulebooks of convolutions with `indice_key=None` are ALSO cached, keyed on (indices, geometry) — the
    reference rebuilds them for 16 of its 17 SubM convs;
  * one fused kernel per convolution instead of gather/mm/scatter-add per offset.
"""
import math

import numpy as np
import torch
from torch.nn import init
from torch.nn.parameter import Parameter

from ..registry import register_everywhere
from . import functional as Fsp
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


def _calculate_fan_in_and_fan_out_hwio(tensor):
    """(fan_in, fan_out) of a filter stored [k..., Cin, Cout] (conv.py:26-42 of the reference; its weights are HWIO, so the
    channel axes are the LAST two and everything in front of them is the receptive field)."""
    if tensor.dim() < 2:
        raise ValueError("fan in and fan out can not be computed for tensor with fewer than 2 dimensions")
    *taps, cin, cout = tensor.shape
    field = 1
    for t in taps:
        field *= int(t)
    return int(cin) * field, int(cout) * field


class IndiceData:
    """What the reference stores in `indice_dict[key]`: the 5-tuple (outids, indices, indice_pairs,
    indice_pair_num, spatial_shape) (conv.py:176-182).  Unpacks like that tuple; the pair arrays are only
    materialised if somebody actually reads them."""

    def __init__(self, rulebook, in_indices, in_spatial_shape):
        self.rulebook = rulebook
        self.in_indices = in_indices
        self.in_spatial_shape = in_spatial_shape
        self.inverse = None   # rulebook of the coupled inverse convolution, built on first use

    def _tuple(self):
        pairs, num = self.rulebook.indice_pairs()
        return (self.rulebook.out_indices, self.in_indices, pairs, num, self.in_spatial_shape)

    def __iter__(self):
        return iter(self._tuple())

    def __getitem__(self, i):
        return self._tuple()[i]

    def __len__(self):
        return 5


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False):
        super().__init__()
        assert groups == 1
        if not isinstance(kernel_size, (list, tuple)):
            kernel_size = [kernel_size] * ndim
        if not isinstance(stride, (list, tuple)):
            stride = [stride] * ndim
        if not isinstance(padding, (list, tuple)):
            padding = [padding] * ndim
        if not isinstance(dilation, (list, tuple)):
            dilation = [dilation] * ndim
        if not isinstance(output_padding, (list, tuple)):
            output_padding = [output_padding] * ndim
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."

        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = list(kernel_size)
        self.conv1x1 = np.prod(kernel_size) == 1
        self.stride = list(stride)
        self.padding = list(padding)
        self.dilation = list(dilation)
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = list(output_padding)
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn

        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = _calculate_fan_in_and_fan_out_hwio(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    # -- rulebook lookup / construction (conv.py:152-183) ----------------------------------------------
    def _geometry_key(self, input):
        ind = input.indices
        return ("__geom__", ind.data_ptr(), ind.shape[0], tuple(input.spatial_shape), tuple(self.kernel_size),
                tuple(self.stride), tuple(self.padding), tuple(self.dilation), bool(self.subm), bool(self.transposed),
                tuple(self.output_padding))

    def get_rulebook(self, input):
        if self.ndim not in (2, 3):
            raise NotImplementedError(f"{self.ndim}D sparse convolution is not implemented (2D and 3D are)")
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            # the coupled convolution's pairs with inputs and outputs swapped (conv.py:153-158)
            assert datas is not None and self.indice_key is not None
            rb = datas.rulebook
            assert rb.kernel_volume == int(np.prod(self.kernel_size)), \
                "inverse conv must have same kernel size as its couple conv"
            if datas.inverse is None:
                datas.inverse = ops.inverse_rulebook(rb, datas.in_indices, datas.in_spatial_shape)
            return datas.inverse
        if self.indice_key is not None and datas is not None:
            return datas.rulebook
        gkey = self._geometry_key(input)
        datas = input.indice_dict.get(gkey)
        if datas is None:
            rb = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size, self.stride,
                                    self.padding, self.dilation, self.subm, self.transposed, self.output_padding)
            datas = IndiceData(rb, input.indices, input.spatial_shape)
            input.indice_dict[gkey] = datas
        if self.indice_key is not None:
            input.indice_dict[self.indice_key] = datas
        return datas.rulebook

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features = input.features
        if self.conv1x1:
            features = torch.mm(input.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                features += self.bias
            out_tensor = SparseConvTensor(features, input.indices, input.spatial_shape, input.batch_size)
            out_tensor.indice_dict = input.indice_dict
            out_tensor.grid = input.grid
            return out_tensor
        rb = self.get_rulebook(input)
        if self.fused_bn:
            # conv.py:184-195: bias folded into the convolution call, no autograd (the reference's has none either)
            assert self.bias is not None
            out_features = ops.sparse_conv(features, self.weight, rb.conv_tables()[0], rb.num_out, bias=self.bias)
        else:
            out_features = Fsp.rulebook_conv(features, self.weight, rb)
            if self.bias is not None:
                out_features = out_features + self.bias.to(out_features.dtype)
        out_tensor = SparseConvTensor(out_features, rb.out_indices, rb.out_spatial_shape, input.batch_size)
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseConv4d(SparseConvolution):
    """Constructible (configs and checkpoints keep loading); forward raises — there is no 4D rulebook here."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(4, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseConvTranspose2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         transposed=True, indice_key=indice_key)


class SparseConvTranspose3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         transposed=True, indice_key=indice_key)


class SparseInverseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(2, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SubMConv4d(SparseConvolution):
    """Constructible only, like SparseConv4d."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(4, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


# the reference registers exactly these ten classes in mmcv's CONV_LAYERS (conv.py:226-455)
for _cls in (SparseConv2d, SparseConv3d, SparseConv4d, SparseConvTranspose2d, SparseConvTranspose3d, SparseInverseConv2d,
             SparseInverseConv3d, SubMConv2d, SubMConv3d, SubMConv4d):
    register_everywhere("conv", _cls)
