"""SparseModule / SparseSequential — mirror of `mmdet3d/ops/spconv/modules.py:40-139`."""
import sys
from collections import OrderedDict

import torch
from torch import nn

from .structure import SparseConvTensor


def is_spconv_module(module):
    spconv_modules = (SparseModule,)
    return isinstance(module, spconv_modules)


def is_sparse_conv(module):
    from .conv import SparseConvolution

    return isinstance(module, SparseConvolution)


class SparseModule(nn.Module):
    """Place holder: modules deriving from it receive the SparseConvTensor itself inside SparseSequential."""

    pass


class SparseSequential(SparseModule):
    """Sequential container: spconv modules get the sparse tensor, ordinary nn.Modules get `.features`."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if sys.version_info < (3, 6):
                raise ValueError("kwargs only supported in py36+")
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        for k, module in self._modules.items():
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                input = module(input)
            else:
                if isinstance(input, SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        input.features = module(input.features)
                else:
                    input = module(input)
        return input


class ToDense(SparseModule):
    """SparseConvTensor -> dense NCHW tensor (modules.py ToDense)."""

    def forward(self, x: SparseConvTensor):
        return x.dense()


class RemoveGrid(SparseModule):
    def forward(self, x: SparseConvTensor):
        x.grid = None
        return x
