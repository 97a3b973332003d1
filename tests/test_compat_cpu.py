"""compat.install(): the reference's caller code imports ``model.*`` / ``pcdet.*`` (train_func.py:20-34); after install()
those names must resolve to this package, the reference's own YAML (Multimodal_cfg/mil3.yaml) must load through the
mirrored ``cfg_from_yaml_file`` and ``GAR_Fusion_ALL(cfg, dataset)`` must build from it with the reference's parameter
names.  Runs in a child process (install() edits sys.modules / sys.meta_path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIL3 = "/root/reference/Multimodal_cfg/mil3.yaml"

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import multimodal_gar_amd.compat as compat
compat.install()
from model.gat_model import *                      # noqa: F401,F403  (train_func.py:20)
from model.gat_model import GAR_Fusion_ALL, GAR_Fusion_Net3, FusionAttention_mat, RGB_Backbone, LiDAR_Backbone
from model.backbone import InceptionI3d, NLBlockND, Unit3D
from model.sg_model import SocialGrouping_model
from pcdet.config import cfg, cfg_from_yaml_file
from pcdet.models import build_network, load_data_to_gpu
from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as pb
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as ps, voxel_pool_modules, voxel_query_utils
from pcdet.datasets.processor.data_processor import DataProcessor                 # dataloader.py:10-12
from pcdet.datasets.processor.point_feature_encoder import PointFeatureEncoder
from pcdet.models.backbones_3d.vfe.mean_vfe import MeanVFE
import multimodal_gar_amd.model.gat_model as real
assert GAR_Fusion_ALL is real.GAR_Fusion_ALL
for name in ("ball_query", "grouping_operation", "farthest_point_sample", "furthest_point_sample", "gather_operation", "three_nn",
             "three_interpolate", "QueryAndGroup", "GroupAll"):
    assert hasattr(pb, name), name
for name in ("ball_query", "grouping_operation", "farthest_point_sample", "stack_farthest_point_sample", "three_nn",
             "three_interpolate", "QueryAndGroup"):
    assert hasattr(ps, name), name
assert hasattr(voxel_pool_modules, "NeighborVoxelSAModuleMSG") and hasattr(voxel_query_utils, "voxel_query")
yaml_path = %(yaml)r
if yaml_path:
    import numpy as np
    from multimodal_gar_amd.pcdet.config import EasyDict
    c = cfg_from_yaml_file(yaml_path, cfg)
    assert c.LiDAR_BACKBONE.MODEL.NAME == "VoxelRCNN" and c.GAR_MODEL.FUSION == "Attention_mat"
    assert c.LiDAR_BACKBONE.MODEL.ROI_HEAD.ROI_GRID_POOL.GRID_SIZE == 6
    dp = c.LiDAR_BACKBONE.DATA_CONFIG if "DATA_CONFIG" in c.LiDAR_BACKBONE else None

    class DS:       # what detector3d_template.py:36-44 reads from the dataset
        class_names = c.LiDAR_BACKBONE.CLASS_NAMES
        point_feature_encoder = EasyDict(num_point_features=4)
        point_cloud_range = np.array([-10.0, -10.0, -2.0, 10.0, 10.0, 2.0], np.float32)
        voxel_size = [0.05, 0.05, 0.1]
        grid_size = np.array([400, 400, 40])
        depth_downsample_factor = None
    import torch
    torch.manual_seed(0)
    net = GAR_Fusion_ALL(c, DS())
    names = dict(net.named_parameters())
    n = sum(p.numel() for p in names.values())
    assert 40e6 < n < 50e6, n                                   # round-1 review measured 45.1 M
    for key in ("GAR_model.AttFusModule1.WQ_r", "RGB_backbone.backbone_net.Mixed_4f.b0.conv3d.weight",
                "LiDAR_backbone.model.roi_head.roi_grid_pool_layers.0.mlps_in.0.0.weight"):
        assert key in names, key
    assert tuple(names["GAR_model.AttFusModule1.WQ_r"].shape) == (512, 512)
    print("built GAR_Fusion_ALL from mil3.yaml: %%.1f M parameters" %% (n / 1e6))
    # the loader, configured by the same YAML as train_func.py:502-507 does, on a synthetic JRDB tree
    import tempfile
    sys.path.insert(0, %(root)r + "/tests")
    import jrdb_tree
    from dataloader import JRDB_act                               # train_func.py:20
    import data.utils.jrdb_transforms as jt                       # dataloader.py:9
    from data.utils.utils import load_pointcloud                  # dataloader.py:8
    import multimodal_gar_amd.dataloader as real_loader
    assert JRDB_act is real_loader.JRDB_act and hasattr(jt, "transform_pts_upper_velodyne_to_base")
    with tempfile.TemporaryDirectory() as tmp:
        root, anns = jrdb_tree.make_tree(tmp)
        aug = c.DATALOADER.train.augmentation
        ds = JRDB_act(aug, root, True, jrdb_tree.NUM_ACTIONS, aug.get("train_backbone", False))
        assert list(ds.grid_size) == [2000, 2000, 40] and ds.num_boxes == 100 and ds.num_frames == 15
        s = ds[2]
        assert s[0].shape == (15, 3, 720, 1280) and s[1].shape == (100, 4) and s[3].shape == (100, 7) and s[9].shape == (100, 5)
        assert s[-1]["points"].shape[1] == 4 and len(s[-1]["points"]) <= 35000 and s[-1]["voxels"].shape[1:] == (5, 4)
        batch = ds.collate_batch([s])
        assert batch[0].shape == (1, 15, 3, 720, 1280) and batch[-1]["voxel_coords"].shape[1] == 4
    print("built JRDB_act from mil3.yaml")
print("compat ok")
"""


def _run(yaml_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "yaml": yaml_path}], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "compat ok" in r.stdout
    return r.stdout


def test_compat_install_aliases_reference_import_names():
    _run("")


@pytest.mark.skipif(not os.path.exists(MIL3), reason="the reference tree (and its mil3.yaml) exists in the build container only")
def test_compat_install_builds_gar_fusion_all_from_reference_yaml():
    out = _run(MIL3)
    assert "built GAR_Fusion_ALL from mil3.yaml" in out and "built JRDB_act from mil3.yaml" in out
