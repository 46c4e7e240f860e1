"""Module containers of the sparse-conv package — API of `mmdet3d/ops/spconv/modules.py:40-139` (`SparseModule`,
`SparseSequential`, `ToDense`, `RemoveGrid`): a `SparseSequential` hands the `SparseConvTensor` itself to sparse modules
and only `.features` to ordinary `nn.Module`s (BatchNorm1d, ReLU, ...)."""
from collections import OrderedDict

from torch import nn

from .structure import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: subclasses receive the SparseConvTensor, not its feature matrix."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


def is_sparse_conv(module):
    from .conv import SparseConvolution

    return isinstance(module, SparseConvolution)


class SparseSequential(SparseModule):
    def __init__(self, *modules, **named):
        super().__init__()
        if len(modules) == 1 and isinstance(modules[0], OrderedDict):
            entries = list(modules[0].items())
        else:
            entries = [(str(i), m) for i, m in enumerate(modules)]
        for name, module in entries + list(named.items()):
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self._sparity_dict = {}

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        n = len(self)
        if not -n <= idx < n:
            raise IndexError("index {} is out of range".format(idx))
        return list(self._modules.values())[idx % n]

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
        from . import bn as native_bn

        entries = list(self._modules.items())
        skip = False
        for i, (name, module) in enumerate(entries):
            if skip:            # a ReLU that the BatchNorm in front of it already applied (one fused launch pair)
                skip = False
                continue
            sparse_in = isinstance(input, SparseConvTensor)
            if is_spconv_module(module):
                assert sparse_in, f"{type(module).__name__} needs a SparseConvTensor"
                self._sparity_dict[name] = input.sparity
                input = module(input)
            elif sparse_in:
                if input.indices.shape[0] != 0:     # dense modules see the [N, C] feature matrix; empty sets are skipped
                    if native_bn.usable(module, input.features):
                        # training-mode BatchNorm1d [+ the ReLU that follows] on the HIP kernels (csrc/sparse_bn.hip)
                        relu = i + 1 < len(entries) and type(entries[i + 1][1]) is nn.ReLU
                        input.features = native_bn.bn_act(input.features, module, relu=relu)
                        skip = relu
                    else:
                        input.features = module(input.features)
            else:
                input = module(input)
        return input


class ToDense(SparseModule):
    """SparseConvTensor -> dense [B, C, *spatial]."""

    def forward(self, x: SparseConvTensor):
        return x.dense()


class RemoveGrid(SparseModule):
    def forward(self, x: SparseConvTensor):
        x.grid = None
        return x
