"""SparseEncoder — the VoxelNet sparse 3D backbone with the interface of `mmdet3d/models/backbones/sparse_encoder.py:10-217`.

Constructor kwargs, attribute names, the module tree and therefore every state-dict key (`conv_input.0.weight`,
`encoder_layers.encoder_layer1.0.conv1.weight`, ..., `conv_out.0.weight`) are the reference's, so its configs build
this class unchanged and its checkpoints load by key.  `voxel_features` is cast to half when `fp16_enabled` is set, as
mmcv's `@auto_fp16(apply_to=("voxel_features",))` does.

Two execution paths: in eval mode with 16-bit weights and gradients off, `spconv/fused.py` runs the whole encoder
sync-free (folded BatchNorm, device-side row counts, one gather for the dense tail); otherwise the modules run one by
one, making the same native calls the reference makes.
"""
import os
import warnings

import torch
from torch import nn

from . import spconv
from .registry import register_everywhere
from .sparse_block import SparseBasicBlock, make_sparse_convmodule
from .spconv import fused as _fused
from .spconv import fused_train as _fused_train

_BLOCK_TYPES = ("conv_module", "basicblock")


class SparseEncoder(nn.Module):
    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type="conv_module"):
        super().__init__()
        if block_type not in _BLOCK_TYPES:
            raise AssertionError(f"block_type must be one of {_BLOCK_TYPES}")
        if not (isinstance(order, (list, tuple)) and sorted(order) == ["act", "conv", "norm"]):
            raise AssertionError("order must be a permutation of ('conv', 'norm', 'act')")
        self.in_channels = in_channels
        self.sparse_shape = sparse_shape
        self.order = tuple(order)
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(encoder_channels)
        self.fp16_enabled = False
        # eval-mode 16-bit forward runs on the sync-free fused path (spconv/fused.py); set False (or
        # BEVAMD_SPCONV_FUSED=0) to force the module-by-module path that mirrors the reference call for call
        self.fused_inference = os.environ.get("BEVAMD_SPCONV_FUSED", "1") != "0"
        # training-mode forward with 16-bit compute and rows in linear order runs on the same kernels (spconv/fused_train.py)
        self.fused_training = os.environ.get("BEVAMD_SPCONV_FUSED_TRAIN", "1") != "0"
        # which path the last forward() took: "fused" | "fused-train" | "modules" (+ why, in `last_path_reason`); the module path makes one
        # host sync per strided convolution and cannot be captured into a HIP graph, so a silent demotion is a perf cliff
        self.last_path = None
        self.last_path_reason = None
        self._warned = set()

        # stem: a pre-activation order keeps only the convolution here (sparse_encoder.py:62-80)
        stem_order = self.order if self.order[0] == "conv" else ("conv",)
        self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                 indice_key="subm1", conv_type="SubMConv3d", order=stem_order)
        stage_out = self.make_encoder_layers(make_sparse_convmodule, norm_cfg, base_channels, block_type=block_type)
        # head: collapses z with a (1, 1, 3) / stride (1, 1, 2) convolution (sparse_encoder.py:87-97)
        self.conv_out = make_sparse_convmodule(stage_out, output_channels, kernel_size=(1, 1, 3), stride=(1, 1, 2),
                                               norm_cfg=norm_cfg, padding=0, indice_key="spconv_down2",
                                               conv_type="SparseConv3d")

    # ------------------------------------------------------------------------------------------------------
    def forward(self, voxel_features, coors, batch_size, **kwargs):
        """voxel_features [N, C_in]; coors [N, 4] int32 (batch, x, y, z) -> dense BEV features [B, C*D, H, W]
        (sparse_encoder.py:100-132).  `num_voxels=` (int32 device tensor) marks capacity-padded inputs;
        `coors_order="linear"`: the caller vouches that the rows are in ascending linear index (b, x, y, z) — what
        `voxelize_batch_device(..., order="key")` writes — so level 1 takes the staged-rows kernels and the sorted-key
        neighbour search (the dense output does not depend on the row order)."""
        num_voxels = kwargs.get("num_voxels")
        coors_order = kwargs.get("coors_order")
        geometry = kwargs.get("geometry")      # fused.prepare_geometry(...) of these coordinates, built ahead of time
        reason = "fused_inference is off"
        if self.training and torch.is_grad_enabled() and self.fused_training:
            # training on the inference kernels (spconv/fused_train.py): 16-bit compute, rows promised in linear order
            reason = _fused_train.unsupported_reason(self, voxel_features, coors_order, num_voxels)
            if reason is None:
                try:
                    out = _fused_train.run_encoder(self, voxel_features, coors, int(batch_size), coors_order=coors_order)
                    self.last_path, self.last_path_reason = "fused-train", None
                    return out
                except _fused.NotThisCall as e:
                    reason = f"this call only: {e}"
                except _fused.Unfusable as e:
                    self.fused_training = False
                    reason = f"encoder not fusable for training, fused training path disabled for this module: {e}"
                    self._warn_once(reason)
        elif self.fused_inference:
            reason = _fused.unsupported_reason(self, voxel_features)
            if reason is None:
                try:
                    out = _fused.run_encoder(self, voxel_features, coors, int(batch_size), num_voxels, geometry=geometry,
                                             coors_order=coors_order)
                    self.last_path, self.last_path_reason = "fused", None
                    return out
                except _fused.NotThisCall as e:
                    reason = f"this call only: {e}"          # e.g. an empty frame: module path for this call only
                except _fused.Unfusable as e:
                    self.fused_inference = False             # a module tree / state the fused path does not implement
                    reason = f"encoder not fusable, fused path disabled for this module: {e}"
                    self._warn_once(reason)
            elif not self.training and not torch.is_grad_enabled():
                self._warn_once(reason)                      # inference that silently misses the fast path
        self.last_path, self.last_path_reason = "modules", reason
        if num_voxels is not None:            # the module path needs exact row counts on the host
            live = int(num_voxels.reshape(-1)[0])
            voxel_features, coors = voxel_features[:live], coors[:live]
        if voxel_features.shape[1] != self.in_channels and voxel_features.shape[1] == _fused.ops.padded_channels(self.in_channels):
            voxel_features = voxel_features[:, : self.in_channels]   # the voxelizer's zero-padded encoder rows on the module path
        if self.fp16_enabled and voxel_features.dtype == torch.float32:
            voxel_features = voxel_features.half()
        x = spconv.SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, int(batch_size))
        x = self.conv_input(x)
        for stage in self.encoder_layers:
            x = stage(x)
        dense = self.conv_out(x).dense()                       # [B, C, H, W, D]
        b, c, h, w, d = dense.shape
        return dense.permute(0, 1, 4, 2, 3).contiguous().view(b, c * d, h, w)

    def prepare_geometry(self, coors, batch_size, num_voxels=None, coors_order=None):
        """The coordinate-only part of the fused forward (rulebook chain of every level), to be run ahead of / beside other work:
        `lvl = enc.prepare_geometry(coors, B, num_voxels=cnt); ...; enc(feats, coors, B, num_voxels=cnt, geometry=lvl)`."""
        return _fused.prepare_geometry(self, coors, int(batch_size), num_voxels, coors_order=coors_order)

    def _warn_once(self, reason):
        if reason not in self._warned:
            self._warned.add(reason)
            warnings.warn(f"SparseEncoder: module-by-module path instead of the sync-free fused path ({reason}); one host sync "
                          "per strided convolution, not graph-capturable", RuntimeWarning, stacklevel=3)

    # ------------------------------------------------------------------------------------------------------
    def make_encoder_layers(self, make_block, norm_cfg, in_channels, block_type="conv_module",
                            conv_cfg=dict(type="SubMConv3d")):
        """Stages `encoder_layer{i}` (sparse_encoder.py:134-217).  conv_module: stage i > 1 OPENS with a strided
        SparseConv3d; basicblock: every stage but the last CLOSES with one, the other entries are residual blocks."""
        if block_type not in _BLOCK_TYPES:
            raise AssertionError(f"block_type must be one of {_BLOCK_TYPES}")
        self.encoder_layers = spconv.SparseSequential()
        last_stage = len(self.encoder_channels) - 1
        out_channels = in_channels
        for i, widths in enumerate(self.encoder_channels):
            widths = tuple(widths)
            pads = tuple(self.encoder_paddings[i])
            blocks = []
            for j, out_channels in enumerate(widths):
                pad = tuple(pads[j]) if isinstance(pads[j], list) else pads[j]
                if block_type == "basicblock":
                    downsample = j == len(widths) - 1 and i != last_stage
                else:
                    downsample = j == 0 and i != 0
                if downsample:
                    blocks.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2, padding=pad,
                                             indice_key=f"spconv{i + 1}", conv_type="SparseConv3d"))
                elif block_type == "basicblock":
                    blocks.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg, conv_cfg=conv_cfg))
                else:
                    blocks.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, padding=pad,
                                             indice_key=f"subm{i + 1}", conv_type="SubMConv3d"))
                in_channels = out_channels
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", spconv.SparseSequential(*blocks))
        return out_channels


register_everywhere("backbone", SparseEncoder)
