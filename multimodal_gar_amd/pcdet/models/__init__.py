"""Minimal mirror of the reference's pcdet/models package surface that MGAR-net touches:
``build_network`` (pcdet/models/__init__.py:15-20 -> detectors.build_detector) and
``load_data_to_gpu`` (:23-35).  Only the detector topologies reachable from MGAR-net are
registered; the other 14 OpenPCDet detectors are out of scope."""
import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict):
    """numpy arrays -> device tensors, in place (same key rules as the reference)."""
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray):
            continue
        if key in ['frame_id', 'metadata', 'calib']:
            continue
        if key in ['images']:
            batch_dict[key] = torch.from_numpy(val).float().cuda().contiguous()
        elif key in ['image_shape']:
            batch_dict[key] = torch.from_numpy(val).int().cuda()
        else:
            batch_dict[key] = torch.from_numpy(val).float().cuda()
