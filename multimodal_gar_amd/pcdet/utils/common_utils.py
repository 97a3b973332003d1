"""The pcdet helpers that feed voxel RoI pooling and the input pipeline (reference pcdet/utils/common_utils.py):
rotate_points_along_z (:35-57), mask_points_by_range (:60-63), get_voxel_centers (:66-82), get_pad_params (:120-137),
generate_voxel2pinds (:235-252).
torch / numpy only; the rest of the reference file (loggers, dist init, SharedArray) is outside the
hot path."""
import numpy as np
import torch


def check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def rotate_points_along_z(points, angle):
    """points (B, N, 3 + C), angle (B) about z, x -> y positive."""
    points, is_numpy = check_numpy_to_torch(points)
    angle, _ = check_numpy_to_torch(angle)
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    # geometry is fp32 on every configuration: under the bf16 configurations' autocast this matmul would otherwise run in
    # bf16 and move the RoI grid points (and with them the ball / voxel query indices)
    with torch.autocast(device_type=points.device.type, enabled=False):
        out = torch.cat((torch.matmul(points[:, :, 0:3].float(), rot), points[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out


def mask_points_by_range(points, limit_range):
    """points (N, 3 + C), limit_range [x0, y0, z0, x1, y1, z1] -> (N) bool: inside the x / y range, bounds included (z is not
    tested)."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) \
        & (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


def get_pad_params(desired_size, cur_size):
    """(before, after) for np.pad: everything that is missing goes after."""
    assert desired_size >= cur_size
    return (0, desired_size - cur_size)


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """voxel_coords (N, 3) [z, y, x] -> centres (N, 3) xyz."""
    assert voxel_coords.shape[1] == 3
    centers = voxel_coords[:, [2, 1, 0]].float()
    vs = torch.tensor(voxel_size, device=centers.device).float() * downsample_times
    lo = torch.tensor(point_cloud_range[0:3], device=centers.device).float()
    return (centers + 0.5) * vs + lo


def scatter_point_inds(indices, point_inds, shape):
    ret = -1 * torch.ones(*shape, dtype=point_inds.dtype, device=point_inds.device)
    flat = indices.view(-1, indices.shape[-1])
    ret[tuple(flat[:, i] for i in range(flat.shape[1]))] = point_inds
    return ret


DENSE_VOXEL2PINDS_MAX_CELLS = 1 << 26     # 256 MB of int32: beyond this the lookup structure is a hash table


def generate_voxel2pinds(sparse_tensor, dense=None):
    """Voxel -> row lookup of a sparse tensor.  dense=True: the reference's (B, Z, Y, X) int32 table (row id in each
    occupied cell, -1 elsewhere; common_utils.py:244-252).  dense=False: a sparse_ops.VoxelHash over the same voxels, which
    the voxel query consumes directly (csrc/voxel_query.hip, identical results) -- O(active voxels) instead of 80 MB per
    sample at the shipped grid.  dense=None: the table while it is small, the hash beyond DENSE_VOXEL2PINDS_MAX_CELLS."""
    if dense is None:
        cells = sparse_tensor.batch_size
        for s in sparse_tensor.spatial_shape:
            cells *= int(s)
        dense = cells <= DENSE_VOXEL2PINDS_MAX_CELLS or not sparse_tensor.indices.is_cuda
    if not dense:
        from ...sparse_ops import VoxelHash
        return VoxelHash(sparse_tensor.indices.int().contiguous(), sparse_tensor.spatial_shape)
    indices = sparse_tensor.indices.long()
    rows = torch.arange(indices.shape[0], device=indices.device, dtype=torch.int32)
    return scatter_point_inds(indices, rows, [sparse_tensor.batch_size] + list(sparse_tensor.spatial_shape))
