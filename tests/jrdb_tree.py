"""A small synthetic JRDB-act directory tree (the layout dataloader.py:19-23 reads) for the input-pipeline tests."""
import os

import numpy as np
from PIL import Image

from multimodal_gar_amd.data.utils.utils import write_pcd
from multimodal_gar_amd.pcdet.config import EasyDict

NUM_ACTIONS = 5


def loader_config(image_size=(36, 64), num_boxes=6, num_frames=3, num_points=-1, shuffle=False, max_voxels=400):
    return EasyDict({
        "image_size": list(image_size), "num_boxes": num_boxes, "sample": {"num_frames": num_frames},
        "point_cloud": {"num_points": num_points, "voxel_size": [0.5, 0.5, 1.0]},
        "POINT_CLOUD_RANGE": [-8.0, -8.0, -2.0, 8.0, 8.0, 2.0], "NUM_POINT_FEATURES": 4,
        "POINT_FEATURE_ENCODING": {"encoding_type": "absolute_coordinates_encoding", "used_feature_list": ['x', 'y', 'z', 'intensity'],
                                   "src_feature_list": ['x', 'y', 'z', 'intensity']},
        "DATA_PROCESSOR": [
            {"NAME": "mask_points_and_boxes_outside_range", "REMOVE_OUTSIDE_BOXES": True},
            {"NAME": "shuffle_points", "SHUFFLE_ENABLED": {"train": shuffle, "test": False}},
            {"NAME": "transform_points_to_voxels", "VOXEL_SIZE": [0.5, 0.5, 1.0], "MAX_POINTS_PER_VOXEL": 4,
             "MAX_NUMBER_OF_VOXELS": {"train": max_voxels, "test": max_voxels}},
        ],
    })


def make_tree(root, seed=0, sequences=("bytes-cafe", "clark-center"), frames=(4, 5, 6, 7), in_size=(24, 188), n_upper=700, n_lower=500,
              missing=()):
    """-> (root path with trailing slash, annotation dict).  ``missing``: (sequence name, fid) image files left out."""
    rng = np.random.default_rng(seed)
    base = os.path.join(str(root), "train_dataset_with_activity")
    anns = {}
    for sid, seq in enumerate(sorted(sequences)):
        os.makedirs(os.path.join(base, "images", "image_stitched", seq))
        for vel in ("lower_velodyne", "upper_velodyne"):
            os.makedirs(os.path.join(base, "pointclouds", vel, seq))
        anns[sid] = {}
        for fid in frames:
            if (seq, fid) not in missing:
                # smooth content: JPEG round-trips it stably
                yy, xx = np.mgrid[0:in_size[0], 0:in_size[1]]
                img = np.stack([(xx * 3 + fid * 7) % 256, (yy * 9 + sid * 40) % 256, (xx + yy * 2) % 256], -1).astype(np.uint8)
                Image.fromarray(img).save(os.path.join(base, "images", "image_stitched", seq, "%06d.jpg" % fid), quality=95)
            for vel, n in (("lower_velodyne", n_lower), ("upper_velodyne", n_upper)):
                pts = np.concatenate([rng.uniform(-10, 10, (n, 2)), rng.uniform(-1.5, 1.5, (n, 1)), rng.uniform(0, 1, (n, 1))], 1)
                write_pcd(os.path.join(base, "pointclouds", vel, seq, "%06d.pcd" % fid), pts.astype(np.float32),
                          data="binary" if fid % 2 else "ascii")
            k = int(rng.integers(1, 5))
            acts = [[int(v) for v in rng.integers(0, 2, NUM_ACTIONS)] for _ in range(k)]
            anns[sid][fid] = {
                "bboxes_3d": [dict(cx=float(rng.uniform(-9, 9)), cy=float(rng.uniform(-9, 9)), cz=float(rng.uniform(-1, 1)), l=0.6, w=0.5,
                                   h=1.7, rot_z=float(rng.uniform(-3, 3))) for _ in range(k)],
                "bboxes_2d": [[float(v) for v in (rng.uniform(0, 0.7), rng.uniform(0, 0.5), rng.uniform(0.05, 0.3), rng.uniform(0.1, 0.5))]
                              for _ in range(k)],
                "actions": acts, "social_group_activity": [a[::-1] for a in acts],
                "person_id": [int(v) for v in rng.integers(0, 50, k)], "social_group_id": [int(v) for v in rng.integers(0, 3, k)],
            }
    os.makedirs(os.path.join(base, "labels_2019"))
    for phase in ("train", "test"):
        np.save(os.path.join(base, "labels_2019", "%s_annotations.npy" % phase), anns, allow_pickle=True)
    return str(root) + os.sep, anns
