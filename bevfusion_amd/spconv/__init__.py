"""spconv — MI355X-native mirror of `mmdet3d/ops/spconv` (the subset BEVFusion's SparseEncoder uses:
SubMConv3d, SparseConv3d, SparseConvTensor, SparseSequential, rulebook + conv forward/backward)."""
from .conv import SparseConv3d, SparseConvolution, SubMConv3d
from .modules import SparseModule, SparseSequential, ToDense, RemoveGrid
from .ops import (Rulebook, build_rulebook, get_conv_output_size, get_deconv_output_size, get_indice_pairs,
                  indice_conv, indice_conv_backward, sparse_conv, sparse_conv_ext)
from .structure import SparseConvTensor, scatter_nd

__all__ = ["SparseConv3d", "SubMConv3d", "SparseConvolution", "SparseModule", "SparseSequential", "ToDense",
           "RemoveGrid", "SparseConvTensor", "scatter_nd", "Rulebook", "build_rulebook", "get_indice_pairs",
           "indice_conv", "indice_conv_backward", "sparse_conv", "sparse_conv_ext", "get_conv_output_size",
           "get_deconv_output_size"]
