"""spconv — MI355X-native mirror of `mmdet3d/ops/spconv`: the sparse tensor, the module containers, the 2D/3D
convolution family (regular, submanifold, transposed, inverse), max pooling, and the rulebook + conv/pool
forward/backward ops under their reference names."""
from .conv import (SparseConv2d, SparseConv3d, SparseConv4d, SparseConvolution, SparseConvTranspose2d,
                   SparseConvTranspose3d, SparseInverseConv2d, SparseInverseConv3d, SubMConv2d, SubMConv3d, SubMConv4d)
from .modules import SparseModule, SparseSequential, ToDense, RemoveGrid
from .ops import (Rulebook, build_rulebook, get_conv_output_size, get_deconv_output_size, get_indice_pairs,
                  indice_conv, indice_conv_backward, indice_maxpool, indice_maxpool_backward, sparse_conv,
                  sparse_conv_ext, sparse_maxpool)
from .pool import SparseMaxPool2d, SparseMaxPool3d
from .structure import SparseConvTensor, scatter_nd

__all__ = ["SparseConv2d", "SparseConv3d", "SparseConv4d", "SubMConv2d", "SubMConv3d", "SubMConv4d",
           "SparseConvTranspose2d", "SparseConvTranspose3d", "SparseInverseConv2d", "SparseInverseConv3d",
           "SparseConvolution", "SparseModule", "SparseSequential", "SparseMaxPool2d", "SparseMaxPool3d", "ToDense",
           "RemoveGrid", "SparseConvTensor", "scatter_nd", "Rulebook", "build_rulebook", "get_indice_pairs",
           "indice_conv", "indice_conv_backward", "indice_maxpool", "indice_maxpool_backward", "sparse_conv",
           "sparse_maxpool", "sparse_conv_ext", "get_conv_output_size", "get_deconv_output_size"]
