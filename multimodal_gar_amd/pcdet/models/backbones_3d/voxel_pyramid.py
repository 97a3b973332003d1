"""Stand-in for the spconv trunk of Voxel R-CNN.

The reference's ``VoxelBackBone8x`` (pcdet/models/backbones_3d/spconv_backbone.py:69-170) is built
on spconv 2.2.3 (third-party, CUDA-only wheels, not installed here); a native sparse convolution
is ranked "next" in SURVEY.md section 8f and is NOT part of this round.  Voxel RoI pooling (which IS on
the hot path) needs the trunk's outputs -- sparse feature maps at strides 2 / 4 / 8 with 32 / 64 /
64 channels -- so this module produces tensors of exactly that structure with a cheap, clearly
different computation: average-pool the occupied voxels into the coarser grid, then a point-wise
Linear + BatchNorm + ReLU.  It is a placeholder for inputs, not a re-implementation of spconv,
and makes no parity claim.
"""
import torch
import torch.nn as nn


class SparseTensorLite:
    """The four attributes of spconv.SparseConvTensor that the RoI head reads."""

    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features = features          # (V, C)
        self.indices = indices            # (V, 4) int32 [b, z, y, x]
        self.spatial_shape = list(spatial_shape)  # [Z, Y, X]
        self.batch_size = batch_size


def _coarsen(features, indices, spatial_shape, batch_size, factor):
    z, y, x = [-(-s // factor) for s in spatial_shape]
    idx = indices.long()
    c = torch.stack([idx[:, 0], idx[:, 1] // factor, idx[:, 2] // factor, idx[:, 3] // factor], 1)
    key = ((c[:, 0] * z + c[:, 1]) * y + c[:, 2]) * x + c[:, 3]
    uniq, inv = torch.unique(key, return_inverse=True)
    summed = torch.zeros((uniq.numel(), features.shape[1]), device=features.device, dtype=features.dtype)
    summed.index_add_(0, inv, features)
    cnt = torch.zeros((uniq.numel(), 1), device=features.device, dtype=features.dtype)
    cnt.index_add_(0, inv, torch.ones_like(features[:, :1]))
    xb = uniq // (z * y * x); r = uniq % (z * y * x)
    new_idx = torch.stack([xb, r // (y * x), (r % (y * x)) // x, r % x], 1).int()
    return summed / cnt, new_idx, [z, y, x]


class VoxelPyramidStandIn(nn.Module):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.sparse_shape = [int(grid_size[2]), int(grid_size[1]), int(grid_size[0])]  # [Z, Y, X]
        chans = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}
        self.lifts = nn.ModuleDict()
        prev = input_channels
        for name, c in chans.items():
            self.lifts[name] = nn.Sequential(nn.Linear(prev, c, bias=False), nn.BatchNorm1d(c, eps=1e-3, momentum=0.01),
                                             nn.ReLU())
            prev = c
        self.num_point_features = 64
        self.backbone_channels = chans

    def forward(self, batch_dict):
        feats, coords = batch_dict['voxel_features'], batch_dict['voxel_coords'].int()
        batch_size = batch_dict['batch_size']
        shape = self.sparse_shape
        out, strides = {}, {}
        stride = 1
        for name in ('x_conv1', 'x_conv2', 'x_conv3', 'x_conv4'):
            if name != 'x_conv1':
                feats, coords, shape = _coarsen(feats, coords, shape, batch_size, 2)
                stride *= 2
            feats = self.lifts[name](feats)
            out[name] = SparseTensorLite(feats, coords, shape, batch_size)
            strides[name] = stride
        batch_dict.update({'encoded_spconv_tensor': out['x_conv4'], 'encoded_spconv_tensor_stride': 8,
                           'multi_scale_3d_features': out, 'multi_scale_3d_strides': strides})
        return batch_dict
