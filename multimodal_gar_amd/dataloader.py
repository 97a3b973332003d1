"""JRDB-act clip loader: the interface of the reference's ``dataloader.py`` (``JRDB_act``: constructor :17-78, ``get_frames``
:91-111, ``one_hot`` :113-117, ``load_pc`` :119-131, ``load_samples_sequence`` :133-293, ``collate_batch`` :295-419), plus a
device path for the per-clip arithmetic.

On-disk layout read (dataloader.py:19-23):
    <root>train_dataset_with_activity/labels_2019/{train,test}_annotations.npy     pickled dict  anns[sid][fid] -> dict with
        'bboxes_3d' (list of dicts cx, cy, cz, l, w, h, rot_z), 'bboxes_2d' (list of [x, y, w, h], fractions of the image),
        'actions', 'social_group_activity' (lists of multi-hot lists), 'person_id', 'social_group_id' (lists of int)
    <root>train_dataset_with_activity/images/image_stitched/<sequence>/<fid:06d>.jpg
    <root>train_dataset_with_activity/pointclouds/{lower,upper}_velodyne/<sequence>/<fid:06d>.pcd

Two ways to use it:

* ``JRDB_act(...)`` with ``device_prep=False`` (default) is the reference's loader: every sample leaves ``__getitem__`` as
  finished float32 tensors made on the host (Pillow resize, float32 normalisation, numpy voxeliser) and
  ``collate_batch`` builds the 12-tuple ``GAR_Fusion_ALL.forward`` takes.
* with ``device_prep=True`` a sample carries the decoded uint8 frames and the two raw velodyne clouds; ``collate_batch``
  only stacks them, and ``DeviceClipPrep`` turns the batch into the same 12-tuple ON THE GPU: one resize + normalise launch
  per clip (csrc/input_prep.hip, Pillow-exact), velodyne merge + range crop in one ordered compaction, shuffle and the batched
  voxeliser on the device.  A quarter of the bytes cross PCIe and no float32 frame is touched by the host.

What cannot be pinned: ``data.utils.utils`` / ``data.utils.jrdb_transforms`` are missing from the reference repository, so
``load_pointcloud``, ``get_lidar_with_sweeps`` and the two sensor transforms are restated from the public JRDB toolkit (see
those modules' headers: PARITY UNPINNED).  Differences from the reference that are deliberate:
  * the key-frame cloud is read and processed once per clip, not once per frame (the reference reads the SAME file
    ``num_frames`` times, :179-180, and keeps the last result, :293);
  * ``load_pc`` transforms every point; the reference's ``pc[:3] = f(pc[:3])`` (:125-126) on an (N, 4) array would only touch
    the first three rows -- pass ``literal_transform_rows=True`` to get that;
  * more annotated boxes than ``num_boxes`` raise instead of looping forever (:247).
"""
import os
import random
from collections import defaultdict

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

from .data.utils import jrdb_transforms as jt
from .data.utils.utils import get_lidar_with_sweeps, load_pointcloud
from .pcdet.datasets.processor.data_processor import DataProcessor, points_to_voxels_batch
from .pcdet.datasets.processor.point_feature_encoder import PointFeatureEncoder
from .pcdet.models.backbones_3d.vfe.mean_vfe import MeanVFE
from .pcdet.utils import common_utils

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resize_to_tensor_normalize(img, image_size, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """PIL image -> (3, H, W) float32 tensor: what ``Compose([Resize(image_size), ToTensor(), Normalize(mean, std)])`` returns
    (dataloader.py:47-49; torchvision is not in this image, these are its three documented steps on a PIL input: Pillow's
    bilinear resize to (H, W); uint8 HWC -> float32 CHW / 255; (x - mean) / std in float32)."""
    h, w = int(image_size[0]), int(image_size[1])
    img = img.convert("RGB")
    if img.size != (w, h):
        img = img.resize((w, h), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    m = torch.as_tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.as_tensor(std, dtype=torch.float32).view(3, 1, 1)
    return x.sub_(m).div_(s)


class JRDB_act(data.Dataset):
    def __init__(self, config, root_path, is_train, num_actions, train_backbone, device_prep=False,
                 literal_transform_rows=False):
        phase = 'train' if is_train else 'test'
        self.anns = np.load(root_path + 'train_dataset_with_activity/labels_2019/{}_annotations.npy'.format(phase),
                            allow_pickle=True).item()
        self.frames = self._all_frames(self.anns)                 # (sequence id, key frame id)
        self.image_path = os.path.join(root_path, 'train_dataset_with_activity/images/image_stitched')
        self.pc_path = os.path.join(root_path, 'train_dataset_with_activity/pointclouds/lower_velodyne')
        self.image_size = config.image_size
        self.is_training = True                                   # as the reference: always the training behaviour (:27)
        self.is_finetune = train_backbone
        self.num_actions = num_actions
        self.num_boxes = config.num_boxes
        self.num_frames = config.sample.num_frames
        self.feature_size = (112, 12)
        self._num_points = config.point_cloud.num_points
        vs = config.point_cloud.voxel_size
        voxel_size = np.array(vs, dtype=np.float32) if isinstance(vs, (list, tuple)) else np.array([vs, vs, vs], dtype=np.float32)
        self._voxel_size = voxel_size.reshape(3, 1)
        self.class_names = ['Pedestrian']
        self.device_prep = device_prep
        self.literal_transform_rows = literal_transform_rows
        self.transforms = lambda img: resize_to_tensor_normalize(img, self.image_size)
        self.point_cloud_range = np.array(config.POINT_CLOUD_RANGE, dtype=np.float32)
        self.point_feature_encoder = PointFeatureEncoder(config.POINT_FEATURE_ENCODING, point_cloud_range=self.point_cloud_range)
        self.data_processor = DataProcessor(config.DATA_PROCESSOR, point_cloud_range=self.point_cloud_range,
                                            training=self.is_training,
                                            num_point_features=self.point_feature_encoder.num_point_features)
        self.grid_size = self.data_processor.grid_size
        self.voxel_size = self.data_processor.voxel_size
        self.depth_downsample_factor = getattr(self.data_processor, "depth_downsample_factor", None)
        self.vfe = MeanVFE(config, num_point_features=self.point_feature_encoder.num_point_features,
                           point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size, grid_size=self.grid_size,
                           depth_downsample_factor=self.depth_downsample_factor)
        self._seq_names = None

    def __getitem__(self, index):
        return self.load_samples_sequence(self.get_frames(self.frames[index]))

    def __len__(self):
        return len(self.frames)

    def _all_frames(self, anns):
        return [(s, f) for s in anns for f in anns[s]]

    def get_frames(self, frame):
        """(sid, key fid) -> [(sid, key fid, fid)]: one random frame of the window when fine-tuning the backbone, else the
        ``num_frames`` frames centred on the key frame."""
        sid, src_fid = frame
        if self.is_finetune:
            if self.is_training:
                return [(sid, src_fid, random.randint(src_fid, src_fid + self.num_frames - 1))]
            return [(sid, src_fid, fid) for fid in range(src_fid, src_fid + self.num_frames)]
        half = self.num_frames // 2
        return [(sid, src_fid, fid) for fid in range(src_fid - half, src_fid + half + 1)]

    def one_hot(self, labels, num_categories):
        result = [0 for _ in range(num_categories)]
        for label in labels:
            result[label] = 1
        return result

    # ---- LiDAR ----------------------------------------------------------------------------------------------------
    def load_pc_raw(self, url):
        """-> (upper (Nu, 4), lower (Nl, 4)) float32, each in its own sensor frame."""
        return load_pointcloud(url.replace('lower_velodyne', 'upper_velodyne')), load_pointcloud(url)

    def load_pc(self, urls):
        """Both velodynes in the base frame, upper first, then ``get_lidar_with_sweeps`` -> (num_points, 4)."""
        pc_upper, pc_lower = self.load_pc_raw(urls)
        if self.literal_transform_rows:                           # rows 0..2 read as a (3, 4) block of x / y / z rows
            pc_upper[:3] = jt.transform_pts_upper_velodyne_to_base(pc_upper[:3])
            pc_lower[:3] = jt.transform_pts_lower_velodyne_to_base(pc_lower[:3])
        else:
            pc_upper[:, :3] = jt.transform_pts_upper_velodyne_to_base(pc_upper[:, :3].T).T
            pc_lower[:, :3] = jt.transform_pts_lower_velodyne_to_base(pc_lower[:, :3].T).T
        pc = np.concatenate([pc_upper, pc_lower], axis=0)
        return get_lidar_with_sweeps(pc, self._num_points)

    # ---- one clip -------------------------------------------------------------------------------------------------
    def _sequence_names(self):
        if self._seq_names is None:
            self._seq_names = sorted(os.listdir(self.image_path))
        return self._seq_names

    def _frame_file(self, seq, fid):
        return self.image_path + '/' + seq + '/' + str(fid).zfill(6) + ".jpg"

    def _labels(self, sid, src_fid, frame_ids):
        """The annotation tensors of a clip (dataloader.py:184-292): everything is the KEY frame's annotation, repeated for
        every frame of the clip and zero / -1 padded to ``num_boxes`` rows."""
        ann = self.anns[sid][src_fid]
        h, w = self.image_size[0], self.image_size[1]
        boxes3d = [(b['cx'], b['cy'], b['cz'], b['l'], b['w'], b['h'], b['rot_z']) for b in ann['bboxes_3d']]
        boxes = [(x * w, y * h, (x + bw) * w, (y + bh) * h) for (x, y, bw, bh) in ann['bboxes_2d']]
        n = len(boxes)
        if n > self.num_boxes or len(boxes3d) > self.num_boxes:
            raise ValueError("sequence %r frame %r has %d boxes, num_boxes is %d" % (sid, src_fid, max(n, len(boxes3d)), self.num_boxes))
        pad = self.num_boxes - n
        zero_action = [0 for _ in range(self.num_actions)]
        t = len(frame_ids)
        bboxes = np.zeros((t, self.num_boxes, 4), np.float32)
        bboxes[:, :n] = np.asarray(boxes, np.float32).reshape(n, 4)
        bboxes3d = np.zeros((t, self.num_boxes, 7), np.float32)
        bboxes3d[:, :len(boxes3d)] = np.asarray(boxes3d, np.float32).reshape(len(boxes3d), 7)
        actions = np.asarray(list(ann['actions']) + [zero_action] * pad, np.float32).reshape(self.num_boxes, self.num_actions)
        group_act = np.asarray(list(ann['social_group_activity']) + [zero_action] * pad, np.float32).reshape(self.num_boxes, self.num_actions)
        person_id = np.asarray(list(ann['person_id']) + [-1] * pad, np.int64)
        group_id = np.asarray(list(ann['social_group_id']) + [-1] * pad, np.int64)
        seq_id = np.full((t, self.num_boxes), -1, np.int64)
        frame_id = np.full((t, self.num_boxes), -1, np.int64)
        k = len(ann['person_id'])
        seq_id[:, :k] = sid
        frame_id[:, :k] = np.asarray(frame_ids, np.int64).reshape(t, 1)
        return {
            "bboxes": torch.from_numpy(bboxes[-1]).float(), "bboxes3d": torch.from_numpy(bboxes3d[-1]).float(),
            "bboxes_num": torch.full((t,), n, dtype=torch.int32), "person_id": torch.from_numpy(person_id),
            "social_group_id": torch.from_numpy(group_id), "seq_id": torch.from_numpy(seq_id), "frame_id": torch.from_numpy(frame_id),
            "actions": torch.from_numpy(actions), "social_group_activity": torch.from_numpy(group_act),
            "gt_boxes": np.asarray(boxes3d, np.float32).reshape(len(boxes3d), 7),
        }

    def load_samples_sequence(self, select_frames):
        """-> (images (T, 3, H, W), bboxes (num_boxes, 4), key fid, bboxes3d (num_boxes, 7), bboxes_num (T), person_id,
        social_group_id (num_boxes), seq_id, frame_id (T, num_boxes), actions, social_group_activity (num_boxes, num_actions),
        data_dict) -- or, with ``device_prep``, the raw sample ``DeviceClipPrep`` finishes on the GPU."""
        seq_names = self._sequence_names()
        sid, src_fid, _ = select_frames[0]
        if not os.path.exists(self._frame_file(seq_names[sid], src_fid)):       # no such key frame: fall back to sample 0
            select_frames = self.get_frames(self.frames[0])
            sid, src_fid, _ = select_frames[0]
        seq = seq_names[sid]
        pics = []
        for (_, _, fid) in select_frames:                          # a missing neighbour frame is replaced by the key frame
            path = self._frame_file(seq, fid)
            pics.append(Image.open(path if os.path.exists(path) else self._frame_file(seq, src_fid)))
        lab = self._labels(sid, src_fid, [fid for (_, _, fid) in select_frames])
        pc_url = os.path.join(self.pc_path, seq, str(src_fid).zfill(6) + '.pcd')
        if self.device_prep:
            frames = torch.from_numpy(np.stack([np.asarray(p.convert("RGB"), dtype=np.uint8) for p in pics]))
            upper, lower = self.load_pc_raw(pc_url)
            return {"frames_u8": frames, "upper": torch.from_numpy(upper), "lower": torch.from_numpy(lower), "src_fid": src_fid,
                    "labels": lab}
        images = torch.stack([self.transforms(p) for p in pics]).float()
        data_dict = {'points': torch.from_numpy(self.load_pc(pc_url)), 'gt_boxes': lab["gt_boxes"]}
        data_dict = self.point_feature_encoder.forward(data_dict)
        data_dict = self.data_processor.forward(data_dict=data_dict)
        return (images, lab["bboxes"], src_fid, lab["bboxes3d"], lab["bboxes_num"], lab["person_id"], lab["social_group_id"],
                lab["seq_id"], lab["frame_id"], lab["actions"], lab["social_group_activity"], data_dict)

    # ---- batches --------------------------------------------------------------------------------------------------
    @staticmethod
    def _label_columns(samples):
        """samples: per clip the 11 leading entries of the reference tuple -> the 11 stacked float tensors / lists."""
        col = lambda i: torch.stack([s[i] for s in samples]).float()
        return (col(1), [s[2] for s in samples], col(3), col(4), col(5), col(6), col(7), col(8), col(9), col(10))

    @staticmethod
    def collate_pcdet(dicts):
        """Per-clip pcdet ``data_dict``s -> one batch dict of numpy arrays (the key rules of dataloader.py:323-414)."""
        merged = defaultdict(list)
        for d in dicts:
            for key, val in d.items():
                merged[key].append(val)
        b = len(dicts)
        ret = {}
        for key, val in merged.items():
            if key in ('voxels', 'voxel_num_points', 'point2img'):
                ret[key] = np.concatenate(val, axis=0)
            elif key in ('points', 'voxel_coords', 'bm_points'):    # a leading column with the clip's index in the batch
                ret[key] = np.concatenate([np.pad(np.asarray(v), ((0, 0), (1, 0)), mode='constant', constant_values=i)
                                           for i, v in enumerate(val)], axis=0)
            elif key in ('gt_boxes', 'gt_boxes2d'):
                rows = max(len(v) for v in val)
                out = np.zeros((b, rows, val[0].shape[-1]), dtype=np.float32)
                for k, v in enumerate(val):
                    if len(v):
                        out[k, :len(v)] = v
                ret[key] = out
            elif key in ('images', 'depth_maps', 'overlap_mask', 'depth_mask'):
                hh, ww = max(v.shape[0] for v in val), max(v.shape[1] for v in val)
                padded = []
                for v in val:
                    width = [common_utils.get_pad_params(hh, v.shape[0]), common_utils.get_pad_params(ww, v.shape[1])]
                    width += [(0, 0)] * (v.ndim - 2)
                    padded.append(np.pad(v, pad_width=width, mode='constant', constant_values=0))
                ret[key] = np.stack(padded, axis=0)
            elif key == 'calib':
                ret[key] = val
            elif key == 'points_2d':
                rows = max(len(v) for v in val)
                ret[key] = np.stack([np.pad(v, ((0, rows - len(v)), (0, 0)), mode='constant', constant_values=0) for v in val], axis=0)
            elif key == 'gt_dense':
                continue
            else:
                ret[key] = np.stack(val, axis=0)
        ret['batch_size'] = b
        return ret

    @staticmethod
    def collate_batch(batch_list, _unused=False):
        """Samples -> (rgb (B, T, 3, H, W), bboxes, [key fids], bboxes3d, bboxes_num, person_id, social_group_id, seq_id, frame_id,
        actions, social_group_activity, pcdet batch dict); raw ``device_prep`` samples -> a ``RawClipBatch``."""
        if isinstance(batch_list[0], dict):
            return RawClipBatch(batch_list)
        rgb = torch.stack([s[0] for s in batch_list]).float()
        cols = JRDB_act._label_columns(batch_list)
        return (rgb,) + cols + (JRDB_act.collate_pcdet([s[-1] for s in batch_list]),)


class RawClipBatch(object):
    """What ``collate_batch`` hands over when the dataset runs with ``device_prep``: uint8 frames (B, T, H0, W0, 3) (one tensor
    when every clip has the same size, else a list), the raw clouds, and the finished label tensors."""

    def __init__(self, samples):
        frames = [s["frames_u8"] for s in samples]
        self.frames_u8 = torch.stack(frames) if len({tuple(f.shape) for f in frames}) == 1 else frames
        self.upper = [s["upper"] for s in samples]
        self.lower = [s["lower"] for s in samples]
        self.src_fid = [s["src_fid"] for s in samples]
        self.labels = [s["labels"] for s in samples]

    def __len__(self):
        return len(self.src_fid)

    def pin_memory(self):
        if torch.is_tensor(self.frames_u8):
            self.frames_u8 = self.frames_u8.pin_memory()
        else:
            self.frames_u8 = [f.pin_memory() for f in self.frames_u8]
        self.upper = [u.pin_memory() for u in self.upper]
        self.lower = [u.pin_memory() for u in self.lower]
        return self


class DeviceClipPrep(object):
    """``RawClipBatch`` -> the reference's 12-tuple, computed on ``device`` (see the module docstring).

    ``dataset``: the ``JRDB_act`` the batch came from (sizes, range, processor configuration).
    ``layout``: 'tchw' gives rgb (B, T, 3, H, W) like the reference; 'cthw' gives (B, 3, T, H, W), the I3D trunk's layout.
    ``generator``: a ``torch.Generator`` on the device for the two random steps (cloud sub-sampling, point shuffle).
    """

    def __init__(self, dataset, device="cuda", dtype=torch.float32, layout="tchw", generator=None):
        self.ds = dataset
        self.device = torch.device(device)
        self.dtype = dtype
        self.layout = layout
        self.generator = generator
        self.voxel_cfg = None
        self.shuffle = False
        self.crop = False
        for cfg in dataset.data_processor.data_processor_queue:
            name, c = cfg[0].__name__, cfg[1]
            if name == "transform_points_to_voxels":
                self.voxel_cfg = c
            elif name == "shuffle_points":
                self.shuffle = bool(c.SHUFFLE_ENABLED[dataset.data_processor.mode])
            elif name == "mask_points_and_boxes_outside_range":
                self.crop = True
                self.crop_cfg = c
            elif name not in ("transform_points_to_voxels_placeholder", "calculate_grid_size"):
                raise NotImplementedError("DeviceClipPrep: data processor step %r has no device path" % name)

    def _randperm(self, n):
        return torch.randperm(n, device=self.device, generator=self.generator)

    def _cloud(self, upper, lower):
        from . import input_ops
        ds = self.ds
        upper = upper.to(self.device, non_blocking=True)
        lower = lower.to(self.device, non_blocking=True)
        rng = ds.point_cloud_range
        n_all = upper.shape[0] + lower.shape[0]
        sampling = ds._num_points is not None and ds._num_points > 0 and n_all != ds._num_points and n_all > 0
        open_range = np.array([-np.inf, -np.inf, -np.inf, np.inf, np.inf, np.inf], np.float32)
        fused_crop = self.crop and not sampling                   # the crop follows get_lidar_with_sweeps in the reference
        pc = input_ops.velodyne_merge_crop(upper, lower, jt.rigid_transform("upper"), jt.rigid_transform("lower"),
                                           rng if fused_crop else open_range)
        if sampling:
            k = ds._num_points
            if n_all > k:
                pc = pc[torch.sort(self._randperm(n_all)[:k]).values]
            else:
                extra = torch.randint(0, n_all, (k - n_all,), device=self.device, generator=self.generator) if k - n_all > n_all \
                    else self._randperm(n_all)[:k - n_all]
                pc = torch.cat([pc, pc[extra]], 0)
        enc = ds.point_feature_encoder.forward({"points": pc})
        pc = enc["points"]
        if self.crop and not fused_crop:
            pc = pc[common_utils.mask_points_by_range(pc, torch.from_numpy(rng).to(self.device))]
        if self.shuffle:
            pc = pc[self._randperm(pc.shape[0])]
        return pc.contiguous(), enc["use_lead_xyz"]

    def __call__(self, batch):
        from . import input_ops
        from .pcdet.utils import box_utils
        ds = self.ds
        b = len(batch)
        h, w = int(ds.image_size[0]), int(ds.image_size[1])
        clips = batch.frames_u8 if not torch.is_tensor(batch.frames_u8) else list(batch.frames_u8)
        t = clips[0].shape[0]
        shape = (b, t, 3, h, w) if self.layout == "tchw" else (b, 3, t, h, w)
        rgb = torch.empty(shape, dtype=self.dtype, device=self.device)
        for i, clip in enumerate(clips):
            input_ops.resize_normalize(clip.to(self.device, non_blocking=True), (h, w), dtype=self.dtype, layout=self.layout, out=rgb[i])
        clouds, lead = [], True
        for u, l in zip(batch.upper, batch.lower):
            pc, lead = self._cloud(u, l)
            clouds.append(pc)
        ret = {}
        ret["points"] = torch.cat([torch.cat([pc.new_full((pc.shape[0], 1), float(i)), pc], 1) for i, pc in enumerate(clouds)], 0)
        gts = []
        for lab in batch.labels:
            g = lab["gt_boxes"]
            if self.crop and self.crop_cfg.REMOVE_OUTSIDE_BOXES and ds.data_processor.training and len(g):
                g = g[box_utils.mask_boxes_outside_range_numpy(g, ds.point_cloud_range, min_num_corners=self.crop_cfg.get('min_num_corners', 1),
                                                               use_center_to_filter=self.crop_cfg.get('USE_CENTER_TO_FILTER', True))]
            gts.append(g)
        rows = max(len(g) for g in gts)
        gt = np.zeros((b, rows, 7), np.float32)
        for k, g in enumerate(gts):
            gt[k, :len(g)] = g
        ret["gt_boxes"] = torch.from_numpy(gt).to(self.device)
        if self.voxel_cfg is not None:
            c = self.voxel_cfg
            p = max(pc.shape[0] for pc in clouds)
            far = float(ds.point_cloud_range[3]) + 16.0 * float(c.VOXEL_SIZE[0])       # padding rows fall outside the grid
            stack = torch.full((b, max(p, 1), clouds[0].shape[1]), far, dtype=torch.float32, device=self.device)
            for i, pc in enumerate(clouds):
                stack[i, :pc.shape[0]] = pc
            vox = points_to_voxels_batch(stack, ds.point_cloud_range, c.VOXEL_SIZE, c.MAX_POINTS_PER_VOXEL,
                                         c.MAX_NUMBER_OF_VOXELS[ds.data_processor.mode])
            ret["voxels"] = vox["voxels"] if lead else vox["voxels"][..., 3:]
            ret["voxel_coords"] = vox["voxel_coords"].float()
            ret["voxel_num_points"] = vox["voxel_num_points"]
        ret["use_lead_xyz"] = np.array([lead] * b)
        ret["batch_size"] = b
        dev = lambda key: torch.stack([lab[key] for lab in batch.labels]).float().to(self.device, non_blocking=True)
        return (rgb, dev("bboxes"), list(batch.src_fid), dev("bboxes3d"), dev("bboxes_num"), dev("person_id"), dev("social_group_id"),
                dev("seq_id"), dev("frame_id"), dev("actions"), dev("social_group_activity"), ret)
