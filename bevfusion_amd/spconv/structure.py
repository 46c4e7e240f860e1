"""SparseConvTensor — the sparse activation container of `mmdet3d/ops/spconv/structure.py:21-64` (same constructor,
attribute names and methods, so the reference's modules and configs use it unchanged).

`dense()` is written as one flat index_put on the raveled voxel coordinates (the reference goes through a generic
`scatter_nd` on tuple slices); the fused encoder path never calls it — its dense tail is `bevamd_spconv_dense_bev`."""
import math

import torch


def _ravel(indices, dims):
    """Row-major linear index of integer coordinates [N, len(dims)] (int64)."""
    lin = indices[:, 0].long()
    for axis in range(1, len(dims)):
        lin = lin * int(dims[axis]) + indices[:, axis].long()
    return lin


def scatter_nd(indices, updates, shape):
    """Dense tensor of `shape` with `updates` written at integer `indices` (structure.py:5-18; duplicate indices are not
    combined, as there).  `indices` [..., k] addresses the first k axes of `shape`."""
    k = indices.shape[-1]
    lead, tail = [int(s) for s in shape[:k]], [int(s) for s in shape[k:]]
    flat = updates.new_zeros([math.prod(lead)] + tail)
    flat[_ravel(indices.reshape(-1, k), lead)] = updates.reshape([-1] + tail)
    return flat.view(lead + tail)


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        # features [N, C]; indices [N, 1 + ndim] = (batch, spatial...).  The reference intends an int32 cast here but
        # discards its result (structure.py:31-32); it is applied.  `grid` (a caller-owned dense lookup grid in the
        # reference) is kept for signature parity only: rulebooks are built from hash / rank indices.
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return math.prod(int(s) for s in self.spatial_shape)

    @property
    def sparity(self):
        return self.indices.shape[0] / self.spatial_size / self.batch_size

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    def dense(self, channels_first=True):
        """[B, *spatial, C], or [B, C, *spatial] (contiguous) when channels_first."""
        dims = [int(self.batch_size)] + [int(s) for s in self.spatial_shape]
        out = scatter_nd(self.indices, self.features, dims + [self.features.shape[1]])
        return out.movedim(-1, 1).contiguous() if channels_first else out
