"""Sparse residual block and conv-module factory — mirror of `mmdet3d/ops/sparse_block.py:62-176`.

`SparseBasicBlock` keeps mmdet's BasicBlock parameter layout (`conv1 / bn1 / conv2 / bn2`, properties `norm1 /
norm2`) so reference checkpoints map key-for-key (mmdet 2.20 BasicBlock is un-vendored)."""
from torch import nn

from . import spconv
from .registry import build_conv_layer, build_norm_layer


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None, dilation=1):
        super().__init__()
        norm_cfg = dict(type="BN") if norm_cfg is None else norm_cfg
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        # mmdet BasicBlock: build_conv_layer(conv_cfg, inplanes, planes, 3, stride=, padding=dilation, bias=False)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation,
                                      bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    def forward(self, x):
        from .spconv import bn as native_bn

        identity = x.features
        assert x.features.dim() == 2, f"x.features.dim()={x.features.dim()}"
        out = self.conv1(x)
        if native_bn.usable(self.norm1, out.features):      # training on the GPU: bn + relu as one launch pair (csrc/sparse_bn.hip)
            out.features = native_bn.bn_act(out.features, self.norm1, relu=True)
        else:
            out.features = self.norm1(out.features)
            out.features = self.relu(out.features)
        out = self.conv2(out)
        if self.downsample is not None:
            identity = self.downsample(x)
        if native_bn.usable(self.norm2, out.features, identity):   # bn + identity + relu
            out.features = native_bn.bn_act(out.features, self.norm2, relu=True, residual=identity)
            return out
        out.features = self.norm2(out.features)
        out.features = out.features + identity
        out.features = self.relu(out.features)
        return out


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type="SubMConv3d", norm_cfg=None, order=("conv", "norm", "act")):
    """sparse_block.py:111-176: SparseSequential(conv [, norm] [, ReLU]) in the given order."""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {"conv", "norm", "act"} == {"conv", "norm", "act"}
    conv_cfg = dict(type=conv_type, indice_key=indice_key)
    layers = list()
    for layer in order:
        if layer == "conv":
            if conv_type not in ["SparseInverseConv3d", "SparseInverseConv2d", "SparseInverseConv1d"]:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                               padding=padding, bias=False))
            else:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, bias=False))
        elif layer == "norm":
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == "act":
            layers.append(nn.ReLU(inplace=True))
    return spconv.SparseSequential(*layers)
