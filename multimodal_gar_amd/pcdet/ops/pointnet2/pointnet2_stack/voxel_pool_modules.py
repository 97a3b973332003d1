"""Voxel RoI pooling layer.

Mirror of the reference's pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py
(NeighborVoxelSAModuleMSG): same constructor keywords and sub-module names (groupers,
mlps_in, mlps_pos, mlps_out).
"""
from typing import List

import torch
import torch.nn as nn

from . import voxel_query_utils
from .....nn_utils import PointwiseSequential


class NeighborVoxelSAModuleMSG(nn.Module):
    def __init__(self, *, query_ranges: List[List[int]], radii: List[float],
                 nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(query_ranges) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps_in = nn.ModuleList()
        self.mlps_pos = nn.ModuleList()
        self.mlps_out = nn.ModuleList()
        for max_range, nsample, radius, spec in zip(query_ranges, nsamples, radii, mlps):
            self.groupers.append(voxel_query_utils.VoxelQueryAndGrouping(max_range, radius, nsample))
            self.mlps_in.append(PointwiseSequential(nn.Conv1d(spec[0], spec[1], kernel_size=1, bias=False),
                                              nn.BatchNorm1d(spec[1])))
            self.mlps_pos.append(PointwiseSequential(nn.Conv2d(3, spec[1], kernel_size=1, bias=False),
                                               nn.BatchNorm2d(spec[1])))
            self.mlps_out.append(PointwiseSequential(nn.Conv1d(spec[1], spec[2], kernel_size=1, bias=False),
                                               nn.BatchNorm1d(spec[2]), nn.ReLU()))
        self.relu = nn.ReLU()
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords, features, voxel2point_indices):
        """xyz (N, 3) voxel centres, features (N, C_in), new_xyz (M, 3) grid points,
        new_coords (M, 4) [b, x, y, z] -> (M, sum_k mlps[k][-1]).
        Reference: voxel_pool_modules.py:70-130."""
        new_coords = new_coords[:, [0, 3, 2, 1]].contiguous()  # -> [b, z, y, x]
        per_scale = []
        for k, grouper in enumerate(self.groupers):
            feats_in = self.mlps_in[k](features.permute(1, 0).unsqueeze(0))        # (1, C, N)
            feats_in = feats_in.squeeze(0).permute(1, 0).contiguous()              # (N, C)
            grouped, grouped_xyz, empty = grouper(new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                                  feats_in, voxel2point_indices)
            keep = (~empty).view(-1, 1, 1).to(grouped.dtype)
            grouped = (grouped * keep).permute(1, 0, 2).unsqueeze(0)               # (1, C, M, nsample)
            rel_xyz = ((grouped_xyz - new_xyz.unsqueeze(-1)) * keep).permute(1, 0, 2).unsqueeze(0)
            x = self.relu(grouped + self.mlps_pos[k](rel_xyz))
            if self.pool_method == 'max_pool':
                x = x.max(dim=3).values                                            # (1, C, M)
            elif self.pool_method == 'avg_pool':
                x = x.mean(dim=3)
            else:
                raise NotImplementedError
            per_scale.append(self.mlps_out[k](x).squeeze(0).permute(1, 0))         # (M, C)
        return torch.cat(per_scale, dim=1)
