"""SparseEncoder — mirror of `mmdet3d/models/backbones/sparse_encoder.py:10-217` (VoxelNet sparse 3D backbone).

Same constructor kwargs, module tree and parameter names (`conv_input.0.weight`, `encoder_layers.encoder_layer1.0
.conv1.weight`, ..., `conv_out.0.weight`), so reference configs build it unchanged and checkpoints load by key.
`voxel_features` is cast to half when `fp16_enabled` is set, like mmcv's `@auto_fp16(apply_to=("voxel_features",))`.
"""
import os

import torch
from torch import nn

from . import spconv
from .spconv import fused as _fused
from .registry import register_everywhere
from .sparse_block import SparseBasicBlock, make_sparse_convmodule


class SparseEncoder(nn.Module):
    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type="conv_module"):
        super().__init__()
        assert block_type in ["conv_module", "basicblock"]
        self.sparse_shape = sparse_shape
        self.in_channels = in_channels
        self.order = tuple(order)
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(self.encoder_channels)
        self.fp16_enabled = False
        # eval-mode 16-bit forward runs on the sync-free fused path (spconv/fused.py); set False (or
        # BEVAMD_SPCONV_FUSED=0) to force the module-by-module path that mirrors the reference call for call
        self.fused_inference = os.environ.get("BEVAMD_SPCONV_FUSED", "1") != "0"

        assert isinstance(order, (list, tuple)) and len(order) == 3
        assert set(order) == {"conv", "norm", "act"}

        if self.order[0] != "conv":  # pre activate
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key="subm1", conv_type="SubMConv3d", order=("conv",))
        else:  # post activate
            self.conv_input = make_sparse_convmodule(in_channels, self.base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key="subm1", conv_type="SubMConv3d")

        encoder_out_channels = self.make_encoder_layers(make_sparse_convmodule, norm_cfg, self.base_channels,
                                                        block_type=block_type)

        self.conv_out = make_sparse_convmodule(encoder_out_channels, self.output_channels, kernel_size=(1, 1, 3),
                                               stride=(1, 1, 2), norm_cfg=norm_cfg, padding=0,
                                               indice_key="spconv_down2", conv_type="SparseConv3d")

    def forward(self, voxel_features, coors, batch_size, **kwargs):
        """voxel_features [N, C]; coors [N, 4] int32 (batch_idx, x, y, z) -> [B, C*D, H, W] dense BEV features
        (sparse_encoder.py:100-132)."""
        if self.fused_inference and _fused.encoder_supported(self, voxel_features):
            try:
                return _fused.run_encoder(self, voxel_features, coors, int(batch_size), kwargs.get("num_voxels"))
            except _fused.NotThisCall:
                pass                          # e.g. an empty frame: module path for this call only
            except _fused.Unfusable:
                self.fused_inference = False  # a module tree / state the fused path does not implement
        if kwargs.get("num_voxels") is not None:
            n = int(kwargs["num_voxels"].reshape(-1)[0])  # capacity-padded inputs: the module path needs exact rows
            voxel_features, coors = voxel_features[:n], coors[:n]
        if self.fp16_enabled and voxel_features.dtype == torch.float32:
            voxel_features = voxel_features.half()
        coors = coors.int()
        input_sp_tensor = spconv.SparseConvTensor(voxel_features, coors, self.sparse_shape, int(batch_size))
        x = self.conv_input(input_sp_tensor)

        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)

        out = self.conv_out(encode_features[-1])
        spatial_features = out.dense()

        N, C, H, W, D = spatial_features.shape
        spatial_features = spatial_features.permute(0, 1, 4, 2, 3).contiguous()
        spatial_features = spatial_features.view(N, C * D, H, W)
        return spatial_features

    def make_encoder_layers(self, make_block, norm_cfg, in_channels, block_type="conv_module",
                            conv_cfg=dict(type="SubMConv3d")):
        """sparse_encoder.py:134-217."""
        assert block_type in ["conv_module", "basicblock"]
        self.encoder_layers = spconv.SparseSequential()

        for i, blocks in enumerate(self.encoder_channels):
            blocks_list = []
            for j, out_channels in enumerate(tuple(blocks)):
                padding = tuple(self.encoder_paddings[i])[j]
                if isinstance(padding, list):
                    padding = tuple(padding)
                # each stage started with a spconv layer except the first stage
                if i != 0 and j == 0 and block_type == "conv_module":
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                  padding=padding, indice_key=f"spconv{i + 1}",
                                                  conv_type="SparseConv3d"))
                elif block_type == "basicblock":
                    if j == len(blocks) - 1 and i != len(self.encoder_channels) - 1:
                        blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                      padding=padding, indice_key=f"spconv{i + 1}",
                                                      conv_type="SparseConv3d"))
                    else:
                        blocks_list.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg,
                                                            conv_cfg=conv_cfg))
                else:
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, padding=padding,
                                                  indice_key=f"subm{i + 1}", conv_type="SubMConv3d"))
                in_channels = out_channels
            stage_name = f"encoder_layer{i + 1}"
            stage_layers = spconv.SparseSequential(*blocks_list)
            self.encoder_layers.add_module(stage_name, stage_layers)
        return out_channels


register_everywhere("backbone", SparseEncoder)
