"""MeanVFE: per-voxel mean of its points (reference backbones_3d/vfe/mean_vfe.py:14-32)."""
import torch
import torch.nn as nn


class MeanVFE(nn.Module):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        """voxels (V, max_points, C), voxel_num_points (V) -> voxel_features (V, C)."""
        voxels, counts = batch_dict['voxels'], batch_dict['voxel_num_points']
        denom = torch.clamp_min(counts.view(-1, 1), min=1.0).type_as(voxels)
        batch_dict['voxel_features'] = (voxels.sum(dim=1) / denom).contiguous()
        return batch_dict
