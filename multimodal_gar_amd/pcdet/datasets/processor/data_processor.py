"""Point cloud -> voxels, with the semantics of the voxel generator the reference wraps.

Mirror of the reference's pcdet/datasets/processor/data_processor.py:15-60 (``VoxelGeneratorWrapper``: constructor
keywords and ``generate(points) -> (voxels, coordinates, num_points)``).  The reference delegates to spconv's
``Point2VoxelCPU3d`` (third-party, spconv 2.2.3, not vendored); its published algorithm, restated:

    for every point, in order:   c = floor((p - range_min) / voxel_size) per axis; skip the point if c is outside the grid
        if the voxel c is new:   if max_num_voxels voxels exist already: skip the point;  else append a voxel for c
        if the voxel holds fewer than max_num_points_per_voxel points: append the point;  (else drop it)
    voxels (V, max_points, C) zero padded, coordinates (V, 3) [z, y, x], num_points_per_voxel (V)

i.e. voxels appear in order of their FIRST point, keep their first max_points points, and only the first max_voxels voxels
exist.  ``points_to_voxels_batch`` computes exactly that for a whole batch of clouds at once with sorts and scans on the
tensors' device (no Python loop over points or clouds); the literal loop lives in oracle/oracle.py::voxelize_points_loop
(tests only).
"""
import numpy as np
import torch


def points_to_voxels_batch(points, point_cloud_range, voxel_size, max_points_per_voxel, max_voxels):
    """points (F, P, C) float32 (xyz first) -> dict(voxels (V, max_points, C), voxel_num_points (V) float32,
    voxel_coords (V, 4) int32 [b, z, y, x], voxel_batch_cnt (F) int32); voxels of cloud 0 first, each cloud's voxels in
    order of first appearance."""
    f, p, c = points.shape
    dev = points.device
    lo = torch.as_tensor(np.asarray(point_cloud_range[:3], np.float32), device=dev)
    vs = torch.as_tensor(np.asarray(voxel_size, np.float32), device=dev)
    grid = np.round((np.asarray(point_cloud_range[3:6], np.float64) - np.asarray(point_cloud_range[:3], np.float64))
                    / np.asarray(voxel_size, np.float64)).astype(np.int64)
    gx, gy, gz = int(grid[0]), int(grid[1]), int(grid[2])
    ijk = torch.floor((points[..., :3] - lo) / vs).long()                                  # (F, P, 3) x, y, z cell
    ok = ((ijk >= 0) & (ijk < torch.tensor([gx, gy, gz], device=dev))).all(-1)
    cell = (ijk[..., 2] * gy + ijk[..., 1]) * gx + ijk[..., 0]                              # z-major linear cell id
    ncell = gx * gy * gz
    frame = torch.arange(f, device=dev).view(f, 1).expand(f, p)
    key = torch.where(ok, frame * ncell + cell, torch.full_like(cell, f * ncell)).reshape(-1)   # invalid points sort last
    order = torch.argsort(key, stable=True)                  # stable: inside one voxel the points stay in input order
    skey = key[order]
    n_valid = int(ok.sum().item())
    order, skey = order[:n_valid], skey[:n_valid]
    if n_valid == 0:
        return {"batch_size": f, "voxels": points.new_zeros((0, max_points_per_voxel, c)), "voxel_num_points": points.new_zeros((0,)),
                "voxel_coords": torch.zeros((0, 4), dtype=torch.int32, device=dev),
                "voxel_batch_cnt": torch.zeros((f,), dtype=torch.int32, device=dev)}
    new = torch.ones_like(skey, dtype=torch.bool)
    new[1:] = skey[1:] != skey[:-1]
    seg = torch.cumsum(new, 0) - 1                           # voxel id (in key order) of every sorted point
    start = torch.nonzero(new).view(-1)                      # first sorted position of every voxel
    vkey = skey[start]
    first_pt = order[start]                                  # flat index (frame * P + p) of the voxel's first point
    # order of first appearance: flat index grows with the frame, so one sort orders frames and, inside, first points
    appear = torch.argsort(first_pt)
    rank_of = torch.empty_like(appear)
    rank_of[appear] = torch.arange(appear.numel(), device=dev)
    vframe = vkey // ncell
    per_frame = torch.bincount(vframe, minlength=f)
    frame_start = torch.cumsum(per_frame, 0) - per_frame
    rank_in_frame = rank_of - frame_start[vframe]
    keep_v = rank_in_frame < max_voxels                      # the first max_voxels voxels of every cloud
    # compact the kept voxels in appearance order
    kept_sorted = keep_v[appear]
    new_row = torch.cumsum(kept_sorted, 0) - 1
    row_of_v = torch.full_like(rank_of, -1)
    row_of_v[appear] = torch.where(kept_sorted, new_row, torch.full_like(new_row, -1))
    nv = int(kept_sorted.sum().item())
    pos = torch.arange(n_valid, device=dev) - start[seg]     # rank of the point inside its voxel
    prow = row_of_v[seg]
    keep_p = (prow >= 0) & (pos < max_points_per_voxel)
    voxels = points.new_zeros((nv, max_points_per_voxel, c))
    voxels[prow[keep_p], pos[keep_p]] = points.reshape(-1, c)[order[keep_p]]
    counts = torch.bincount(prow[keep_p], minlength=nv)
    kv = vkey[appear][kept_sorted]
    r = kv % ncell
    coords = torch.stack([kv // ncell, r // (gy * gx), (r % (gy * gx)) // gx, r % gx], 1).int()
    return {"batch_size": f, "voxels": voxels, "voxel_num_points": counts.to(points.dtype), "voxel_coords": coords,
            "voxel_batch_cnt": torch.bincount(coords[:, 0].long(), minlength=f).int()}


class VoxelGeneratorWrapper:
    """Same constructor keywords and ``generate`` contract as the reference's wrapper (data_processor.py:15-60):
    points (N, C) numpy array (or tensor) -> (voxels (V, max_points, C), coordinates (V, 3) [z, y, x], num_points (V)),
    numpy in -> numpy out."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        self.vsize_xyz = [float(v) for v in vsize_xyz]
        self.coors_range_xyz = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_num_points_per_voxel = int(max_num_points_per_voxel)
        self.max_num_voxels = int(max_num_voxels)

    def generate(self, points):
        is_numpy = isinstance(points, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)) if is_numpy else points.float()
        out = points_to_voxels_batch(t.unsqueeze(0), self.coors_range_xyz, self.vsize_xyz, self.max_num_points_per_voxel,
                                     self.max_num_voxels)
        voxels, coords, num = out["voxels"], out["voxel_coords"][:, 1:].contiguous(), out["voxel_num_points"].int()
        if is_numpy:
            return voxels.numpy(), coords.numpy(), num.numpy()
        return voxels, coords, num


class DataProcessor(object):
    """Mirror of the reference's ``DataProcessor`` (data_processor.py:63-245): a queue of the steps the config names, applied
    to one frame's ``data_dict`` on the host.  Steps the shipped config uses (Multimodal_cfg/mil3.yaml:47-59):
    mask_points_and_boxes_outside_range, shuffle_points, transform_points_to_voxels; also sample_points,
    transform_points_to_voxels_placeholder and calculate_grid_size.  (The device path of the same steps for a whole batch:
    multimodal_gar_amd/dataloader.py::DeviceClipPrep.)"""

    def __init__(self, processor_configs, point_cloud_range, training, num_point_features):
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.training = training
        self.num_point_features = num_point_features
        self.mode = 'train' if training else 'test'
        self.grid_size = self.voxel_size = None
        self.voxel_generator = None
        self.data_processor_queue = []
        for cur_cfg in processor_configs:
            step = getattr(self, cur_cfg.NAME)
            step(None, cur_cfg)                                   # the configuration pass (grid size)
            self.data_processor_queue.append((step, cur_cfg))

    def _grid(self, config):
        grid = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(config.VOXEL_SIZE)
        self.grid_size = np.round(grid).astype(np.int64)
        self.voxel_size = config.VOXEL_SIZE

    def mask_points_and_boxes_outside_range(self, data_dict=None, config=None):
        if data_dict is None:
            return None
        from ...utils import box_utils, common_utils
        if data_dict.get('points', None) is not None:
            mask = common_utils.mask_points_by_range(data_dict['points'], self.point_cloud_range)
            data_dict['points'] = data_dict['points'][mask]
        if data_dict.get('gt_boxes', None) is not None and config.REMOVE_OUTSIDE_BOXES and self.training:
            mask = box_utils.mask_boxes_outside_range_numpy(
                data_dict['gt_boxes'], self.point_cloud_range, min_num_corners=config.get('min_num_corners', 1),
                use_center_to_filter=config.get('USE_CENTER_TO_FILTER', True))
            data_dict['gt_boxes'] = data_dict['gt_boxes'][mask]
        return data_dict

    def shuffle_points(self, data_dict=None, config=None):
        if data_dict is None:
            return None
        if config.SHUFFLE_ENABLED[self.mode]:
            points = data_dict['points']
            data_dict['points'] = points[np.random.permutation(points.shape[0])]
        return data_dict

    def transform_points_to_voxels_placeholder(self, data_dict=None, config=None):
        if data_dict is None:
            return self._grid(config)
        return data_dict

    def calculate_grid_size(self, data_dict=None, config=None):
        if data_dict is None:
            return self._grid(config)
        return data_dict

    def transform_points_to_voxels(self, data_dict=None, config=None):
        if data_dict is None:
            return self._grid(config)
        if self.voxel_generator is None:
            self.voxel_generator = VoxelGeneratorWrapper(
                vsize_xyz=config.VOXEL_SIZE, coors_range_xyz=self.point_cloud_range, num_point_features=self.num_point_features,
                max_num_points_per_voxel=config.MAX_POINTS_PER_VOXEL, max_num_voxels=config.MAX_NUMBER_OF_VOXELS[self.mode])
        voxels, coordinates, num_points = self.voxel_generator.generate(data_dict['points'])
        if not data_dict['use_lead_xyz']:
            voxels = voxels[..., 3:]
        data_dict['voxels'] = voxels
        data_dict['voxel_coords'] = coordinates
        data_dict['voxel_num_points'] = num_points
        return data_dict

    def sample_points(self, data_dict=None, config=None):
        if data_dict is None:
            return None
        num_points = config.NUM_POINTS[self.mode]
        if num_points == -1:
            return data_dict
        points = data_dict['points']
        if num_points < len(points):
            near = np.linalg.norm(points[:, 0:3], axis=1) < 40.0
            far_idx, near_idx = np.where(near == 0)[0], np.where(near == 1)[0]
            if num_points > len(far_idx):                         # every far point, the rest drawn from the near ones
                near_choice = np.random.choice(near_idx, num_points - len(far_idx), replace=False)
                choice = np.concatenate((near_choice, far_idx), axis=0) if len(far_idx) > 0 else near_choice
            else:
                choice = np.random.choice(np.arange(0, len(points), dtype=np.int32), num_points, replace=False)
            np.random.shuffle(choice)
        else:
            choice = np.arange(0, len(points), dtype=np.int32)
            if num_points > len(points):
                choice = np.concatenate((choice, np.random.choice(choice, num_points - len(points), replace=False)), axis=0)
            np.random.shuffle(choice)
        data_dict['points'] = points[choice]
        return data_dict

    def forward(self, data_dict):
        for step, cfg in self.data_processor_queue:
            data_dict = step(data_dict, cfg)
        return data_dict
