"""points_in_boxes_gpu of the reference's pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py:28-43 (the RoI-aware voxel
pooling of the same file is detector-training code outside MGAR-net's path)."""
import torch

from . import roiaware_pool3d_cuda


def points_in_boxes_gpu(points, boxes):
    """points (B, M, 3), boxes (B, T, 7) -> box_idxs_of_pts (B, M) int32: index of the first box containing each point,
    background = -1."""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    batch_size, num_points, _ = points.shape
    box_idxs_of_pts = points.new_zeros((batch_size, num_points), dtype=torch.int).fill_(-1)
    roiaware_pool3d_cuda.points_in_boxes_gpu(boxes.contiguous(), points.contiguous(), box_idxs_of_pts)
    return box_idxs_of_pts
