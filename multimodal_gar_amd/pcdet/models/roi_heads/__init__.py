from .voxelrcnn_head import VoxelRCNNHead
from .point_grid_head import PointGridRoIHead

__all__ = {'VoxelRCNNHead': VoxelRCNNHead, 'PointGridRoIHead': PointGridRoIHead}
