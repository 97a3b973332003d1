"""Sparse 3-D trunk of Voxel R-CNN.

Mirror of the reference's pcdet/models/backbones_3d/spconv_backbone.py:8-27 (post_act_block) and :69-170
(VoxelBackBone8x): same constructor, sub-module names and parameter shapes (state dicts load), same outputs
(``encoded_spconv_tensor``, ``multi_scale_3d_features`` x_conv1..4 with strides 1, 2, 4, 8).  The reference builds it on
the third-party spconv; here the ``spconv`` namespace is pcdet/utils/spconv_utils.py on the gfx950 sparse-convolution
kernels of csrc/sparse_conv.hip.
"""
from functools import partial

import torch.nn as nn

from ...utils.spconv_utils import spconv


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm', norm_fn=None):
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)   # 'inverseconv' is not used by VoxelBackBone8x
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class VoxelBackBone8x(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        gs = [int(v) for v in grid_size]
        self.sparse_shape = [gs[2] + 1, gs[1], gs[0]]            # grid_size[::-1] + [1, 0, 0]   (spconv_backbone.py:75)
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        last_pad = self.model_cfg.get('last_pad', 0) if hasattr(self.model_cfg, 'get') else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False, indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}

    def forward(self, batch_dict):
        """voxel_features (V, C), voxel_coords (V, 4) [b, z, y, x] -> encoded_spconv_tensor + multi-scale features."""
        feats, coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        x = spconv.SparseConvTensor(features=feats, indices=coords.int(), spatial_shape=self.sparse_shape,
                                    batch_size=batch_dict['batch_size'])
        x = self.conv_input(x)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        out = self.conv_out(x_conv4)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4}})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict
