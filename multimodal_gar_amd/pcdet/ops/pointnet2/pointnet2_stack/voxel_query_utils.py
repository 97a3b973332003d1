"""Neighbour-voxel query + grouping (stacked layout).

Mirror of the reference's pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py
(VoxelQuery / voxel_query / VoxelQueryAndGrouping), on top of mgar_voxel_query_stack.
"""
from typing import List

import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2
from . import pointnet2_utils


class VoxelQuery(Function):
    """(idx (M, nsample) int32 GLOBAL voxel rows, empty_ball_mask (M) bool).
    Reference: voxel_query_utils.py:10-46 -> voxel_query_gpu.cu:10-89."""

    @staticmethod
    def forward(ctx, max_range: List[int], radius: float, nsample: int, xyz: torch.Tensor,
                new_xyz: torch.Tensor, new_coords: torch.Tensor, point_indices: torch.Tensor):
        assert new_xyz.is_contiguous() and xyz.is_contiguous() and new_coords.is_contiguous()
        n_query = new_coords.shape[0]
        idx = torch.zeros((n_query, nsample), dtype=torch.int32, device=xyz.device)
        _query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices, idx)
        empty_ball_mask = idx[:, 0] == -1
        idx[empty_ball_mask] = 0
        ctx.mark_non_differentiable(idx, empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None, None


def _query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices, idx):
    """point_indices: the reference's dense (B, Z, Y, X) int32 table, or a sparse_ops.VoxelHash over the same voxels."""
    z_range, y_range, x_range = max_range
    n_query = new_coords.shape[0]
    if torch.is_tensor(point_indices):
        assert point_indices.is_contiguous()
        _, gz, gy, gx = point_indices.shape
        pointnet2.voxel_query_wrapper(n_query, gz, gy, gx, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                                      point_indices, idx)
    else:
        gz, gy, gx = point_indices.spatial_shape
        pointnet2.voxel_query_hash_wrapper(n_query, gz, gy, gx, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                                           point_indices, idx)


voxel_query = VoxelQuery.apply


def voxel_query_raw(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """The kernel's own output: idx (M, nsample) int32 GLOBAL voxel rows with idx[m, 0] == -1 for an empty neighbourhood
    (voxel_query_gpu.cu:10-89), without the mask / zero-fill post-processing of VoxelQuery -- what the fused pooling
    kernel consumes."""
    n_query = new_coords.shape[0]
    idx = torch.zeros((n_query, nsample), dtype=torch.int32, device=xyz.device)
    _query(max_range, radius, nsample, xyz.contiguous(), new_xyz.contiguous(), new_coords.contiguous(), point_indices, idx)
    return idx


class VoxelQueryAndGrouping(nn.Module):
    """voxel_query -> make indices sample-local -> group xyz and features.
    Returns (grouped_features (M, C, nsample), grouped_xyz (M, 3, nsample), empty_ball_mask).
    Reference: voxel_query_utils.py:51-100 (requires the same number of queries per sample,
    like the reference's ``idx.view(batch_size, -1, nsample)``)."""

    def __init__(self, max_range: List[int], radius: float, nsample: int):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords: torch.Tensor, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor,
                new_xyz: torch.Tensor, new_xyz_batch_cnt: torch.Tensor,
                features: torch.Tensor, voxel2point_indices: torch.Tensor):
        assert xyz.shape[0] == xyz_batch_cnt.sum(), \
            'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_coords.shape[0] == new_xyz_batch_cnt.sum(), \
            'new_coords: %s, new_xyz_batch_cnt: %s' % (str(new_coords.shape), str(new_xyz_batch_cnt))
        batch_size = xyz_batch_cnt.shape[0]
        idx, empty = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords,
                                 voxel2point_indices)
        # global voxel row -> row inside its own sample (no host sync: offsets stay on device)
        starts = torch.cumsum(xyz_batch_cnt, 0, dtype=torch.int32) - xyz_batch_cnt.int()
        idx = (idx.view(batch_size, -1, self.nsample) - starts.view(-1, 1, 1)).view(-1, self.nsample)
        idx = idx.masked_fill(empty.view(-1, 1), 0).contiguous()
        grouped_xyz = pointnet2_utils.grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_features = pointnet2_utils.grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty
