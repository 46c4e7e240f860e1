"""SparseConvTensor — mirror of `mmdet3d/ops/spconv/structure.py:21-64` (same attributes and methods)."""
import numpy as np
import torch


def scatter_nd(indices, updates, shape):
    """structure.py:5-18: dense tensor with `updates` written at `indices` (no duplicate handling)."""
    ret = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    ndim = indices.shape[-1]
    output_shape = list(indices.shape[:-1]) + shape[indices.shape[-1]:]
    flatted_indices = indices.view(-1, ndim)
    slices = [flatted_indices[:, i] for i in range(ndim)]
    slices += [Ellipsis]
    ret[tuple(slices)] = updates.view(*output_shape)
    return ret


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        """features [N, C]; indices [N, 1+ndim] int32 (batch, spatial...); `grid` is accepted for signature
        parity (the reference's optional pre-allocated dense lookup grid; the hash-based rulebook needs none)."""
        self.features = features
        self.indices = indices
        if self.indices.dtype != torch.int32:
            self.indices = self.indices.int()  # the reference's line (structure.py:31-32) discards the cast
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        if key in self.indice_dict:
            return self.indice_dict[key]
        return None

    def dense(self, channels_first=True):
        output_shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        res = scatter_nd(self.indices.long(), self.features, output_shape)
        if not channels_first:
            return res
        ndim = len(self.spatial_shape)
        trans_params = list(range(0, ndim + 1))
        trans_params.insert(1, ndim + 1)
        return res.permute(*trans_params).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
