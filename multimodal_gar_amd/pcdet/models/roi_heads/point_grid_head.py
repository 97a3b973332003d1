"""Per-actor lift on the PointNet++ route: RoI-grid pooling with ball queries.

BASELINE.json's north-star names the classic set-abstraction stack (ball query, grouping, FPS,
three-NN interpolate) as the LiDAR lift; the reference reaches those ops through the same
registry (pcdet/models/backbones_3d/__init__.py:8-18 -> PointNet2MSG) but ships no YAML that
instantiates them (SURVEY.md "facts").  This head is the composition used for that route: the
PV-RCNN style grid pooling of the reference's pcdet/models/roi_heads/pvrcnn_head.py:64-108
(StackSAModuleMSG on a GRID_SIZE^3 lattice inside every actor box) applied to the per-point
features of PointNet2MSG, with the pooling geometry of mil3.yaml:105-134 (grid 6, nsample 16,
radii 0.4 / 0.8 / 1.6, 32-channel MLPs -> 96 channels) so its output feeds LiDAR_Backbone
exactly like the voxel route's ``pooled_features``.
"""
import torch
import torch.nn as nn

from ...ops.pointnet2.pointnet2_stack import pointnet2_modules as pointnet2_stack_modules
from .voxelrcnn_head import global_grid_points_of_roi


class PointGridRoIHead(nn.Module):
    def __init__(self, input_channels, model_cfg, num_class=1, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        pool = model_cfg.ROI_GRID_POOL
        self.grid_size = pool.GRID_SIZE
        mlps = [[input_channels] + list(m) for m in pool.MLPS]
        self.roi_grid_pool_layer = pointnet2_stack_modules.StackSAModuleMSG(
            radii=pool.POOL_RADIUS, nsamples=pool.NSAMPLE, mlps=mlps, use_xyz=True, pool_method=pool.POOL_METHOD)
        self.num_pooled_channels = sum(m[-1] for m in mlps)

    def forward(self, batch_dict):
        """gt_boxes (B, N, 7), point_coords (P, 4) [b, x, y, z], point_features (P, C)
        -> pooled_features (B*N, G^3, C_out)."""
        rois = batch_dict['gt_boxes']
        batch_size = batch_dict['batch_size']
        g = self.grid_size
        grid_xyz, _ = global_grid_points_of_roi(rois, g)                      # (B*N, G^3, 3)
        new_xyz = grid_xyz.view(-1, 3).contiguous()
        per_sample = rois.shape[1] * g ** 3
        new_cnt = torch.full((batch_size,), per_sample, dtype=torch.int32, device=rois.device)
        coords = batch_dict['point_coords']
        xyz = coords[:, 1:4].contiguous()
        xyz_cnt = batch_dict.get('point_batch_cnt')
        if xyz_cnt is None:
            xyz_cnt = torch.bincount(coords[:, 0].long(), minlength=batch_size).int()
        feats = batch_dict.get('point_features_cm') if xyz.is_cuda else None   # (B, C, n) channel-major, device path
        if feats is None or feats.shape[0] * feats.shape[2] != xyz.shape[0]:
            feats = batch_dict['point_features'].contiguous()                  # (P, C) stacked rows (reference layout)
        _, pooled = self.roi_grid_pool_layer(xyz=xyz, xyz_batch_cnt=xyz_cnt, new_xyz=new_xyz, new_xyz_batch_cnt=new_cnt,
                                             features=feats)
        batch_dict['pooled_features'] = pooled.reshape(-1, g ** 3, pooled.shape[-1])
        return batch_dict
