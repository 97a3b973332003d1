"""Detector assemblies reachable from MGAR-net.  ``build_detector`` mirrors the reference's
pcdet/models/detectors/__init__.py (name -> class registry)."""
import torch.nn as nn

from ..backbones_3d import __all__ as _backbones
from ..backbones_3d.vfe import __all__ as _vfes
from ..roi_heads import __all__ as _heads


class VoxelRCNN(nn.Module):
    """MeanVFE -> 3D trunk -> VoxelRCNNHead; forward just chains the modules and returns the dict
    (reference pcdet/models/detectors/voxel_rcnn.py:9-13; loss / post-processing are unreachable)."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        nfeat = dataset.point_feature_encoder.num_point_features
        self.vfe = _vfes[model_cfg.VFE.NAME](model_cfg=model_cfg.VFE, num_point_features=nfeat)
        self.backbone_3d = _backbones[model_cfg.BACKBONE_3D.NAME](model_cfg=model_cfg.BACKBONE_3D,
                                                                  input_channels=self.vfe.get_output_feature_dim(),
                                                                  grid_size=dataset.grid_size)
        self.roi_head = _heads[model_cfg.ROI_HEAD.NAME](backbone_channels=self.backbone_3d.backbone_channels,
                                                        model_cfg=model_cfg.ROI_HEAD,
                                                        point_cloud_range=dataset.point_cloud_range,
                                                        voxel_size=dataset.voxel_size, num_class=num_class)
        self.module_list = [self.vfe, self.backbone_3d, self.roi_head]

    def forward(self, batch_dict):
        for m in self.module_list:
            batch_dict = m(batch_dict)
        return batch_dict


class PointNet2RoI(nn.Module):
    """PointNet2MSG (SA x k + FP x k) -> PointGridRoIHead: the set-abstraction route of the
    north-star (see roi_heads/point_grid_head.py)."""

    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        nfeat = dataset.point_feature_encoder.num_point_features
        self.backbone_3d = _backbones[model_cfg.BACKBONE_3D.NAME](model_cfg=model_cfg.BACKBONE_3D, input_channels=nfeat)
        self.roi_head = _heads[model_cfg.ROI_HEAD.NAME](input_channels=self.backbone_3d.num_point_features,
                                                        model_cfg=model_cfg.ROI_HEAD, num_class=num_class)
        self.module_list = [self.backbone_3d, self.roi_head]

    def forward(self, batch_dict):
        for m in self.module_list:
            batch_dict = m(batch_dict)
        return batch_dict


__all__ = {'VoxelRCNN': VoxelRCNN, 'PointNet2RoI': PointNet2RoI}


def build_detector(model_cfg, num_class, dataset):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
