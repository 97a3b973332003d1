from .pointnet2_backbone import PointNet2MSG
from .voxel_pyramid import SparseTensorLite, VoxelPyramidStandIn

__all__ = {
    'PointNet2MSG': PointNet2MSG,
    # spconv's VoxelBackBone8x is a third-party sparse-conv trunk (out of scope, SURVEY.md section 8f
    # rank 1); the name resolves to a documented stand-in with the same outputs' structure.
    'VoxelBackBone8x': VoxelPyramidStandIn,
    'VoxelPyramidStandIn': VoxelPyramidStandIn,
}
