from .pointnet2_backbone import PointNet2MSG
from .spconv_backbone import VoxelBackBone8x

__all__ = {
    'PointNet2MSG': PointNet2MSG,
    # the reference's sparse trunk (spconv_backbone.py:69-170) on this repo's sparse-convolution kernels
    'VoxelBackBone8x': VoxelBackBone8x,
}
