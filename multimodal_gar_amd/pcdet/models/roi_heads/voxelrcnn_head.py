"""Voxel RoI pooling head as MGAR-net uses it.

Mirror of the reference's (locally edited) pcdet/models/roi_heads/voxelrcnn_head.py: RoIs are the
ground-truth actor boxes (:92), each box is sampled on a GRID_SIZE^3 lattice (:167-188), and every
feature source x_conv2/3/4 is pooled with a NeighborVoxelSAModuleMSG (:126-158).  ``forward`` stores
``pooled_features`` (B*N, G^3, C) and ``shared_feature`` and returns the dict; proposal / target /
loss code of OpenPCDet is unreachable in the reference (:196-203 commented out) and is not provided.
"""
import torch
import torch.nn as nn

from ...ops.pointnet2.pointnet2_stack import voxel_pool_modules as voxelpool_stack_modules
from ...utils import common_utils


def dense_grid_points(rois, grid_size):
    """(R, G^3, 3) lattice points in each box's local frame: ((i + .5)/G - .5) * size."""
    r = rois.shape[0]
    ii = torch.arange(grid_size, device=rois.device, dtype=rois.dtype)
    gx, gy, gz = torch.meshgrid(ii, ii, ii, indexing='ij')            # x slowest .. z fastest, as nonzero() orders
    dense_idx = torch.stack([gx, gy, gz], -1).reshape(1, -1, 3).repeat(r, 1, 1)
    size = rois.view(r, -1)[:, 3:6].unsqueeze(1)
    return (dense_idx + 0.5) / grid_size * size - size / 2


def global_grid_points_of_roi(rois, grid_size):
    rois = rois.view(-1, rois.shape[-1])
    local = dense_grid_points(rois, grid_size)
    glob = common_utils.rotate_points_along_z(local.clone(), rois[:, 6]).squeeze(dim=1)
    return glob + rois[:, 0:3].unsqueeze(dim=1), local


class VoxelRCNNHead(nn.Module):
    def __init__(self, backbone_channels, model_cfg, point_cloud_range, voxel_size, num_class=1, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.pool_cfg = model_cfg.ROI_GRID_POOL
        self.point_cloud_range = point_cloud_range
        self.voxel_size = voxel_size
        layers = self.pool_cfg.POOL_LAYERS
        c_out = 0
        self.roi_grid_pool_layers = nn.ModuleList()
        for src in self.pool_cfg.FEATURES_SOURCE:
            mlps = [[backbone_channels[src]] + list(m) for m in layers[src].MLPS]
            self.roi_grid_pool_layers.append(voxelpool_stack_modules.NeighborVoxelSAModuleMSG(
                query_ranges=layers[src].QUERY_RANGES, nsamples=layers[src].NSAMPLE, radii=layers[src].POOL_RADIUS,
                mlps=mlps, pool_method=layers[src].POOL_METHOD))
            c_out += sum(m[-1] for m in mlps)
        g = self.pool_cfg.GRID_SIZE
        pre = g * g * g * c_out
        fc = []
        n_fc = len(model_cfg.SHARED_FC)
        for k in range(n_fc):
            fc += [nn.Linear(pre, model_cfg.SHARED_FC[k], bias=False), nn.BatchNorm1d(model_cfg.SHARED_FC[k]),
                   nn.ReLU(inplace=True)]
            pre = model_cfg.SHARED_FC[k]
            if k != n_fc - 1 and model_cfg.DP_RATIO > 0:
                fc.append(nn.Dropout(model_cfg.DP_RATIO))
        self.shared_fc_layer = nn.Sequential(*fc)
        for m in self.shared_fc_layer.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight)

    get_global_grid_points_of_roi = staticmethod(global_grid_points_of_roi)
    get_dense_grid_points = staticmethod(lambda rois, batch_size_rcnn, grid_size: dense_grid_points(rois, grid_size))

    def roi_grid_pool(self, batch_dict):
        rois = batch_dict['gt_boxes']
        batch_size = batch_dict['batch_size']
        g = self.pool_cfg.GRID_SIZE
        grid_xyz, _ = global_grid_points_of_roi(rois, g)
        grid_xyz = grid_xyz.view(batch_size, -1, 3)
        lo = self.point_cloud_range
        vs = self.voxel_size
        coords = torch.cat([(grid_xyz[:, :, i:i + 1] - lo[i]) // vs[i] for i in range(3)], dim=-1)   # x, y, z voxel ids
        batch_idx = torch.arange(batch_size, device=rois.device, dtype=rois.dtype).view(-1, 1, 1).expand(-1, coords.shape[1], 1)
        grid_cnt = torch.full((batch_size,), coords.shape[1], dtype=torch.int32, device=rois.device)
        pooled = []
        for k, src in enumerate(self.pool_cfg.FEATURES_SOURCE):
            stride = batch_dict['multi_scale_3d_strides'][src]
            sp = batch_dict['multi_scale_3d_features'][src]
            centres = common_utils.get_voxel_centers(sp.indices[:, 1:4], downsample_times=stride, voxel_size=vs,
                                                     point_cloud_range=lo)
            cnt = torch.bincount(sp.indices[:, 0].long(), minlength=batch_size).int()
            v2p = common_utils.generate_voxel2pinds(sp)
            cur = torch.cat([batch_idx, coords // stride], dim=-1).int()                               # [b, x, y, z]
            feat = self.roi_grid_pool_layers[k](xyz=centres.contiguous(), xyz_batch_cnt=cnt,
                                                new_xyz=grid_xyz.contiguous().view(-1, 3), new_xyz_batch_cnt=grid_cnt,
                                                new_coords=cur.contiguous().view(-1, 4), features=sp.features.contiguous(),
                                                voxel2point_indices=v2p)
            pooled.append(feat.view(-1, g ** 3, feat.shape[-1]))
        return torch.cat(pooled, dim=-1)

    def forward(self, batch_dict):
        pooled = self.roi_grid_pool(batch_dict)                           # (B*N, G^3, C)
        batch_dict['pooled_features'] = pooled
        batch_dict['shared_feature'] = self.shared_fc_layer(pooled.view(pooled.size(0), -1))
        return batch_dict
