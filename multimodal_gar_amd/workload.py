"""The benchmark / smoke workload: one MGAR-net training step over a batch of synthetic clips.

The reference has no benchmark; BASELINE.json defines the shape (config c3: 8 clips x 15 frames x
32 actors x 16 384 points, fp32, forward + backward).  What one "clip" is here (frozen builder's
choice, DESIGN.md section "workload"):

  RGB   : the clip's T frames (B, T, 3, H, W) go through I3D once per clip, batch 1, exactly as the
          reference runs it (BATCH_SIZE 1, mil3.yaml:161); centre temporal slice -> RoIAlign of the
          clip's A actor boxes -> non-local block -> Linear -> GATv2 over the actor graph:
          R tokens (A, 512) per clip.
  LiDAR : every one of the clip's T frames is an independent scene of P points with its own A
          actor boxes; all B*T frames form the batch of the PointNet++ stack (SA x 4 + FP x 4:
          FPS, ball query, grouping, three-NN interpolation) and of the per-actor RoI-grid lift
          (6^3 grid points per actor, 3 radii) -> (B*T*A, 216, 96) -> non-local block 3D ->
          Linear(20736, 512): L tokens (A, 512) per frame.
  fuse  : each frame is one scene for GAR_Fusion_Net3 (DAFM x 2, similarity, adjacency, group
          pooling, 14 heads + cardinality) with R of its clip and L of the frame: B*T scenes.
  loss  : a synthetic scalar over all 16 outputs (the reference's JRDB losses need labels and are
          out of scope); backward through everything trainable (I3D is frozen,
          as in the reference) and one Adam step.

Data parallelism: the clip batch is sharded across ranks (one process per GPU); gradients are
all-reduced by DistributedDataParallel over RCCL.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import synthetic as S
from .pcdet.config import EasyDict

PC_RANGE = [-20.0, -20.0, -2.0, 20.0, 20.0, 2.0]


def lidar_model_cfg(n_points, route="pointnet2"):
    if route == "pointnet2":
        p = n_points
        return EasyDict(
            NAME="PointNet2RoI",
            BACKBONE_3D=dict(
                NAME="PointNet2MSG",
                # OpenPCDet's stock PointNet2MSG plan, with the level sizes tied to the cloud size
                SA_CONFIG=dict(NPOINTS=[p // 4, p // 16, p // 64, p // 256],
                               RADIUS=[[0.1, 0.5], [0.5, 1.0], [1.0, 2.0], [2.0, 4.0]],
                               NSAMPLE=[[16, 32], [16, 32], [16, 32], [16, 32]],
                               MLPS=[[[16, 16, 32], [32, 32, 64]], [[64, 64, 128], [64, 96, 128]],
                                     [[128, 196, 256], [128, 196, 256]], [[256, 256, 512], [256, 384, 512]]]),
                FP_MLPS=[[128, 128], [256, 256], [512, 512], [512, 512]],
                # nothing downstream of the trunk reads the stacked (P, C) copy on the device: the RoI head takes the
                # channel-major tensor (point_features_cm), so the ~1 GB transposed copy is skipped
                STACKED_POINT_FEATURES=False),
            ROI_HEAD=dict(NAME="PointGridRoIHead",
                          # pooling geometry of mil3.yaml:105-134
                          ROI_GRID_POOL=dict(GRID_SIZE=6, MLPS=[[32, 32], [32, 32], [32, 32]],
                                             POOL_RADIUS=[0.4, 0.8, 1.6], NSAMPLE=[16, 16, 16], POOL_METHOD="max_pool")))
    if route == "voxel":
        layer = lambda r: dict(MLPS=[[32, 32]], QUERY_RANGES=[[4, 4, 4]], POOL_RADIUS=[r], NSAMPLE=[16],  # noqa: E731
                               POOL_METHOD="max_pool")
        return EasyDict(
            NAME="VoxelRCNN", VFE=dict(NAME="MeanVFE"), BACKBONE_3D=dict(NAME="VoxelBackBone8x"),
            ROI_HEAD=dict(NAME="VoxelRCNNHead", CLASS_AGNOSTIC=True, SHARED_FC=[512, 512], DP_RATIO=0.3,
                          ROI_GRID_POOL=dict(FEATURES_SOURCE=["x_conv2", "x_conv3", "x_conv4"], PRE_MLP=True, GRID_SIZE=6,
                                             POOL_LAYERS=dict(x_conv2=layer(0.4), x_conv3=layer(0.8), x_conv4=layer(1.6)))))
    raise ValueError(route)


def model_cfg(n_actors, n_points, gat=True, route="pointnet2"):
    """The shipped configuration (Multimodal_cfg/mil3.yaml) with the synthetic sizes plugged in.
    GAT_module is False in the shipped YAML (:86); the north-star includes the GAT path, so the
    benchmark flips it on (SURVEY.md "facts")."""
    return EasyDict(
        DATALOADER=dict(train=dict(augmentation=dict(num_boxes=n_actors + 1, image_size=[720, 1280], crop_size=5))),
        RGB_BACKBONE=dict(I3D_FREEZE=True, EMBEDDING_DIM=512, INTER_PERSON=False, GAT_module=bool(gat), two_stage_att=False),
        LiDAR_BACKBONE=dict(CLASS_NAMES=["Pedestrian"], MODEL=lidar_model_cfg(n_points, route),
                            SELF_ATT1=dict(USE=True, DIM=3, INTER_PERSON=False), two_stage_att=False),
        GAR_MODEL=dict(MODALITY="Multi", FUSION="Attention_mat", SIGMA=10, FEAT_NORM=True, EUCLIDEAN=True,
                       ind_action_concat=True, sg_feat_org=False, FEATURE_DIM=1024, HIDDEN_DIM=512, sim="cosine"))


class SyntheticDataset:
    """The attributes pcdet's detector template reads from a dataset (detector3d_template.py:36-44)."""

    def __init__(self, voxel_size=(0.25, 0.25, 0.1)):
        # 160 x 160 x 40 cells over the 40 m x 40 m x 4 m range: 40 cells in z like the shipped grid (mil3.yaml:40,56), which is
        # what VoxelBackBone8x's four z-halvings assume (41 -> 21 -> 11 -> 5 -> 2, spconv_backbone.py:85-117)
        self.class_names = ["Pedestrian"]
        self.point_feature_encoder = EasyDict(num_point_features=4)
        self.point_cloud_range = np.array(PC_RANGE, np.float32)
        self.voxel_size = list(voxel_size)
        self.grid_size = np.round((self.point_cloud_range[3:] - self.point_cloud_range[:3]) / np.array(voxel_size)).astype(np.int64)
        self.depth_downsample_factor = None


def make_batch(seed, n_clips, n_frames, n_actors, n_points, height, width, device):
    """Synthetic batch on `device`: dict of tensors (see module docstring for the meaning)."""
    sc = S.scene_batch(seed, n_clips * n_frames, n_actors, n_points, num_boxes=n_actors + 1, height=height, width=width)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    lab = S.scene_labels(seed + 3, n_clips * n_frames, n_actors, num_boxes=n_actors + 1)
    rng = np.random.default_rng(seed + 1)
    boxes2d = np.zeros((n_clips, n_actors + 1, 4), np.float32)
    for b in range(n_clips):
        boxes2d[b, :n_actors] = S.actor_boxes2d(rng, n_actors, height, width)
    return {
        "images": t(S.images(seed + 2, n_clips, n_frames, height, width)),      # (B, T, 3, H, W)
        "bboxes": t(boxes2d),                                                     # (B, A+1, 4) key-frame boxes
        "points": t(sc["points"]),                                                # (B*T, P, 4)
        "bboxes3d": t(sc["bboxes3d"]),                                            # (B*T, A+1, 7)
        "person_id": t(sc["person_id"]),                                          # (B*T, A+1)
        "social_group_id": t(lab["social_group_id"]),                             # (B*T, A+1) int64, -1 padded
        "action": t(lab["action"]),                                               # (B*T, A+1, 27)
        "social_group_activity": t(lab["social_group_activity"]),                 # (B*T, A+1, 27)
        "n_clips": n_clips, "n_frames": n_frames, "n_actors": n_actors,
    }


class ClipModel(nn.Module):
    """GAR_Fusion_ALL plus the clip/frame plumbing described in the module docstring."""

    def __init__(self, n_actors, n_points, gat=True, route="pointnet2"):
        super().__init__()
        from .model.gat_model import GAR_Fusion_ALL
        self.cfg = model_cfg(n_actors, n_points, gat, route)
        self.route = route
        self.n_actors = n_actors
        self.dataset = SyntheticDataset()
        self.net = GAR_Fusion_ALL(self.cfg, self.dataset)
        self.net.GAR_model.uniform_actor_count = n_actors
        self.overlap_branches = True
        self.batch_i3d = True
        # Opt-in (bench.py --i3d-channels-last): several clips per I3D pass with the activations NDHWC between the stem and
        # the RoI crop, so that MIOpen's convolutions need no layout adapters (model/backbone.py,
        # InceptionI3d.set_channels_last).  Measured at c3: library convolutions 62.5 -> 52.9 ms, this library's BatchNorm /
        # pooling kernels +3.5 ms in that layout, step 229.4 -> ~226 ms -- but MIOpen's search over its NDHWC kernel
        # instances takes 4 min 20 s at start-up on a fresh box (NCDHW: 35-40 s), shipped find-db or not, so it is off.
        self.i3d_channels_last = False
        self._side_stream = self._geo_stream = None
        # What of the trunk's coordinate-only work is issued ahead of the feature path: "fps1" = the level-1 FPS, on the main
        # stream before the I3D launches (round 1); "all" = every level's FPS, ball queries and 3-NN weights on a third
        # stream.  "all" measured SLOWER inside the HIP graph (1 clip: 45.4 vs 42.3 ms; c3: 244.2 vs 235.2 ms/step,
        # profiles/README.md round 2): the extra branch delays the feature path's kernels more than it hides.
        self.geometry_ahead = "fps1"
        # Opt-in (bench.py --prefetch-geometry): input-side software pipelining.  The trunk's coordinate-only work (FPS of all
        # levels, ball queries, 3-NN weights) depends on the points alone, so a step can compute it for the NEXT batch on a side
        # stream while it runs the feature path and the backward of the current one -- as a loader would.  Every step still does
        # the geometry of one batch and the forward + backward of one batch; only the order changes.  It takes the level-1 FPS
        # (4.6 ms on 15 workgroups at one clip per rank) off the critical path.  Off by default: the headline numbers are the
        # un-pipelined step.
        self.geometry_prefetch = False
        self.geometry_stream_max_clouds = int(os.environ.get("MGAR_GEOMETRY_STREAM_MAX_CLOUDS", 60))   # see forward()
        # Two-stream step: the LiDAR branch on a third stream instead of the issuing one (see forward())
        self.branches_off_origin = os.environ.get("MGAR_BRANCHES_OFF_ORIGIN", "1") != "0"
        # Microseconds by which the RGB side stream starts after the level-1 FPS kernel has gone out (see forward()); 0 = off
        self.sampling_head_start_us = int(os.environ.get("MGAR_SAMPLING_HEAD_START_US", 20))
        # Opt-in (bench.py --prefetch-rgb): the same pipelining for the FROZEN RGB branch.  I3D + RoIAlign carry no gradient and
        # depend on the frames alone, so a step can run them for the NEXT batch on the side stream under its own BACKWARD (an
        # MFMA-bound pass beside streaming kernels) instead of beside the LiDAR forward.  Every step still runs one I3D pass; only
        # the order across the step boundary changes.  Off by default, like geometry_prefetch.
        self.rgb_prefetch = False
        self._rgb_cur = None        # RoI crops of the batch this step consumes
        self._rgb_next = None       # crops being computed for the next step (owned by the side stream until finish_prefetch)
        self._geo_cur = None        # geometry of the batch this step consumes (computed during the previous step)
        self._geo_next = None       # geometry being computed for the next step (owned by the side stream until finish_prefetch)

    # ---- RGB: one I3D pass per clip (batch 1, like the reference) ---------------------------------
    def rgb_crops(self, images, bboxes):
        """Frozen part of the RGB branch (I3D_FREEZE): I3D + RoIAlign, no autograd graph.  On the device all clips go
        through I3D in ONE pass with per-clip BatchNorm statistics (= the reference's one pass per clip, batch 1;
        fewer, larger launches: 13.5 instead of 15.0 ms per clip); elsewhere clip by clip."""
        rb = self.net.RGB_backbone
        b = images.shape[0]
        with torch.no_grad():
            if images.is_cuda and b > 1 and self.batch_i3d:
                _B, _T, _C, _H, _W = images.shape
                if self.i3d_channels_last != bool(getattr(rb.backbone_net, "channels_last", False)):
                    rb.backbone_net.set_channels_last(self.i3d_channels_last)
                rb.backbone_net.set_per_sample_stats(True)
                try:
                    crops = rb.crop_features(images.view(_B, _C, _T, _H, _W), [bboxes[i] for i in range(b)])
                finally:
                    rb.backbone_net.set_per_sample_stats(False)
                per = bboxes.shape[1]
                return [crops[i * per:(i + 1) * per] for i in range(b)]
            crops = []
            for i in range(b):
                clip = images[i:i + 1]
                _B, _T, _C, _H, _W = clip.shape
                clip = clip.view(_B, _C, _T, _H, _W)             # the reference's view (gat_model.py:1836)
                crops.append(rb.crop_features(clip, [bboxes[i]]))  # (A+1, 832, 5, 5)
        return crops

    def rgb_tokens_from_crops(self, crops):
        """Trainable tail: non-local block, pooling, embedding, optional GAT -> (B, A, 512)."""
        from .model.gat_model import fully_connected_edges
        rb = self.net.RGB_backbone
        a = self.n_actors
        toks = []
        for c in crops:
            tok = rb.embed(c[:a])                                # (A, 512)
            if rb.cfg.GAT_module:
                tok = rb.GAT_module(tok, fully_connected_edges([a], tok.device))
            toks.append(tok)
        return torch.stack(toks)

    def rgb_tokens(self, images, bboxes):
        return self.rgb_tokens_from_crops(self.rgb_crops(images, bboxes))

    # ---- LiDAR: all frames of all clips in one batch ------------------------------------------------
    def trunk_geometry(self, points, stream=None):
        """The coordinate-only part of the PointNet++ trunk (FPS centres of the four levels, ball queries of the folded
        scales, 3-NN weights of the decoder), issued ahead of the feature path on ``stream``: level-1 FPS runs one workgroup
        per cloud for ~5 ms (6-47 % of the CUs) and overlaps the I3D work (see ``geometry_ahead``)."""
        if self.route != "pointnet2":
            return None
        trunk = self.net.LiDAR_backbone.model.backbone_3d
        if self.geometry_ahead == "all":
            return trunk.geometry(points, stream)
        return trunk.geometry(points, stream, levels=1, balls=False, neighbours=False)

    def lidar_tokens(self, points, bboxes3d, geometry=None):
        f, p, _ = points.shape
        a = self.n_actors
        lb = self.net.LiDAR_backbone
        if self.route == "pointnet2":
            bidx = torch.arange(f, device=points.device, dtype=points.dtype).view(f, 1, 1).expand(f, p, 1)
            data = {"batch_size": f, "points": torch.cat([bidx, points], -1).view(f * p, 5),
                    "gt_boxes": bboxes3d[:, :a, :].contiguous(),
                    "point_batch_cnt": torch.full((f,), p, dtype=torch.int32, device=points.device)}
            if geometry is not None:
                data["trunk_geometry"] = geometry
        else:
            data = voxelize_batch(points, self.dataset)
            data["gt_boxes"] = bboxes3d[:, :a, :].contiguous()
        tok = lb(data)                                           # (1, F*A, 512)
        return tok.view(f, a, -1)

    def finish_prefetch(self):
        """After the step's BACKWARD (its saved index tensors are the current geometry): join the side stream that computed
        the next batch's geometry and make it the current one -- by copying into the current buffers, so that a captured HIP
        graph keeps reading the same addresses."""
        if self._rgb_next is not None:
            main = torch.cuda.current_stream()
            main.wait_stream(self._side_stream)
            capturing = torch.cuda.is_current_stream_capturing()
            for a_, b_ in zip(self._rgb_cur, self._rgb_next):
                if not capturing:
                    b_.record_stream(main)
                a_.copy_(b_)
            self._rgb_next = None
        if self._geo_next is None:
            return
        main = torch.cuda.current_stream()
        main.wait_stream(self._geo_stream)
        src, dst = _geometry_tensors(self._geo_next), _geometry_tensors(self._geo_cur)
        assert len(src) == len(dst)
        capturing = torch.cuda.is_current_stream_capturing()
        for a_, b_ in zip(dst, src):
            if not capturing:
                b_.record_stream(main)
            a_.copy_(b_)
        self._geo_next = None

    def forward(self, batch):
        b, t, a = batch["n_clips"], batch["n_frames"], self.n_actors
        if batch["images"].is_cuda and self.overlap_branches:
            # The frozen I3D pass (MFMA-heavy convolutions, no autograd graph) runs on a side HIP stream next to the
            # LiDAR branch on the main stream.  Issue order: level-1 FPS first (a single launch that leaves most of
            # the chip idle), then the I3D launches, then the rest of the LiDAR branch.  Nothing that autograd will
            # replay lives on the side stream, so gradient hooks (DDP) only ever see the main stream.
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream, self._geo_stream = torch.cuda.Stream(), torch.cuda.Stream()
            inputs_ready = main.record_event()
            self._geo_stream.wait_event(inputs_ready)
            prefetch = self.geometry_prefetch and self.route == "pointnet2"
            if prefetch:
                trunk = self.net.LiDAR_backbone.model.backbone_3d
                if self._geo_cur is None:      # first step: nothing was prefetched -- compute it in line, once
                    self._geo_cur = _without_events(trunk.geometry(batch["points"]))
                geometry = self._geo_cur
                # the NEXT batch's geometry (the caller passes its points; the benchmark's batches are all the same tensor)
                self._geo_next = trunk.geometry(batch.get("next_points", batch["points"]), self._geo_stream)
            else:
                geometry = None
            from . import _lib as L
            from .pcdet.ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as shim
            sampling_goes_out = []
            if not prefetch and self.sampling_head_start_us > 0:
                # The level-1 FPS (one 1024-thread workgroup per cloud, the head of the LiDAR chain) has to be RESIDENT before the
                # I3D stem's workgroups start streaming through the CUs, or it starts when the stem ends (csrc/errors.hip,
                # mgar_delay_us): the side stream waits for the moment the sampling kernel goes out, plus a few microseconds.
                shim.BEFORE_SAMPLING_LAUNCH = lambda: sampling_goes_out.append(torch.cuda.current_stream().record_event())
            try:
                if not prefetch and not (self.branches_off_origin and not self.rgb_prefetch):
                    geometry = self.trunk_geometry(batch["points"], self._geo_stream if self.geometry_ahead == "all" else main)   # FPS first
            finally:
                shim.BEFORE_SAMPLING_LAUNCH = None
            self._side_stream.wait_event(inputs_ready)
            if self.branches_off_origin and not self.rgb_prefetch and not prefetch:
                # Both branches away from the stream the step is issued (and captured) on: the RGB branch on the side stream, the
                # LiDAR branch -- forward here, hence its backward too -- on a third one; the issuing stream only forks and joins.
                # Measured against the LiDAR branch on the issuing stream (same box, 3 runs each): 177.6-178.4 vs 182.1-182.6 ms
                # at 8 clips, 95.6 vs 96.9 at 4, 54.2 vs 55.2 at 2, 34.4 both at 1.
                lst = self._geo_stream
                with torch.cuda.stream(self._side_stream):     # (issued first: with the LiDAR branch first -- and the RGB branch
                    crops = self.rgb_crops(batch["images"], batch["bboxes"])   # held back until its sampling kernel is out -- 179.5 ms)
                    # the trainable RGB tail (non-local block, embedding, GAT) too: its forward beside the LiDAR forward, its
                    # backward beside the LiDAR backward (8 clips 176.3 -> 175.6 ms, 1 clip 32.9 -> 32.5)
                    rgb = self.rgb_tokens_from_crops(crops)
                geo = None
                if self.route == "pointnet2" and batch["points"].shape[0] <= self.geometry_stream_max_clouds:
                    # few clouds: the trunk's coordinate-only chain (FPS of the four levels, ball queries, 3-NN weights -- every
                    # level's FPS waits for the previous one and occupies one workgroup per cloud) on a stream of its own, ahead of
                    # the feature path.  Same box: 33.1 vs 34.1 ms at 1 clip (15 clouds), 53.2 vs 54.1 at 2, 95.1 vs 95.7 at 4;
                    # 178.3 vs 177.5 at 8 (120 clouds: off).
                    if getattr(self, "_geo4_stream", None) is None:
                        self._geo4_stream = torch.cuda.Stream()
                    self._geo4_stream.wait_event(inputs_ready)
                    ga, self.geometry_ahead = self.geometry_ahead, "all"
                    try:
                        geo = self.trunk_geometry(batch["points"], self._geo4_stream)
                    finally:
                        self.geometry_ahead = ga
                with torch.cuda.stream(lst):
                    lidar = self.lidar_tokens(batch["points"], batch["bboxes3d"], geo)
                if geo is not None:
                    main.wait_stream(self._geo4_stream)
                main.wait_stream(self._side_stream)
                main.wait_stream(lst)
                if not torch.cuda.is_current_stream_capturing():
                    for c in crops:
                        c.record_stream(main)
                    lidar.record_stream(main)
                    rgb.record_stream(main)
                return self._fuse(batch, rgb, lidar)
            if self.rgb_prefetch:
                if self._rgb_cur is None:      # first step: nothing was prefetched -- compute it in line, once
                    self._rgb_cur = [c.clone() for c in self.rgb_crops(batch["images"], batch["bboxes"])]
                crops = self._rgb_cur
            else:
                if sampling_goes_out:
                    self._side_stream.wait_event(sampling_goes_out[0])
                    L.call("mgar_delay_us", int(self.sampling_head_start_us), self._side_stream.cuda_stream)
                with torch.cuda.stream(self._side_stream):
                    crops = self.rgb_crops(batch["images"], batch["bboxes"])
            lidar = self.lidar_tokens(batch["points"], batch["bboxes3d"], geometry)   # (B*T, A, 512)
            main.wait_stream(self._side_stream)
            if not prefetch:
                main.wait_stream(self._geo_stream)
            if not torch.cuda.is_current_stream_capturing():   # inside a graph the pool is private and replays are serial
                for c in crops:
                    c.record_stream(main)
            rgb = self.rgb_tokens_from_crops(crops)                                   # (B, A, 512)
        else:
            rgb = self.rgb_tokens(batch["images"], batch["bboxes"])
            lidar = self.lidar_tokens(batch["points"], batch["bboxes3d"])
        return self._fuse(batch, rgb, lidar)

    def _fuse(self, batch, rgb, lidar):
        b, t, a = batch["n_clips"], batch["n_frames"], self.n_actors
        rgb_s = rgb[:, None].expand(b, t, a, rgb.shape[-1]).reshape(b * t, a, -1)   # every frame-scene of a clip
        pad = lambda x: torch.cat([x, x.new_zeros(x.shape[0], 1, x.shape[2])], 1)    # noqa: E731  -> MNP = A + 1
        bb2 = batch["bboxes"][:, None].expand(b, t, a + 1, 4).reshape(b * t, a + 1, 4)
        # The fusion net works on (A, 512) tokens per scene: launch-bound, not bandwidth-bound.  It runs in fp32 on every
        # configuration (under the bf16 configurations the token producers above are bf16; the tokens are widened here).
        with torch.autocast(device_type=rgb_s.device.type, enabled=False):
            out = self.net.GAR_model(pad(rgb_s.float()), pad(lidar.float()), bb2, batch["bboxes3d"], None, batch["person_id"])
        if batch["images"].is_cuda and self.overlap_branches and self.rgb_prefetch:
            # the NEXT batch's frozen RGB pass (the caller passes its frames; the benchmark's batches are all the same tensor),
            # issued where the forward ends: it runs on the side stream under this step's objective and backward
            main = torch.cuda.current_stream()
            self._side_stream.wait_event(main.record_event())
            with torch.cuda.stream(self._side_stream):
                self._rgb_next = self.rgb_crops(batch.get("next_images", batch["images"]), batch.get("next_bboxes", batch["bboxes"]))
        return out


def _geometry_tensors(geo):
    out = list(geo.centres)
    for idx in geo.ball_idx:
        if isinstance(idx, dict):
            out += [idx[k] for k in sorted(idx)]
        elif torch.is_tensor(idx):
            out.append(idx)
    for i in sorted(geo.nn):
        out += list(geo.nn[i])
    return out


def _without_events(geo):
    """The same TrunkGeometry with no stream events: its producer has been joined already (or ran on the consumer's stream)."""
    geo.centre_events = [None] * len(geo.centre_events)
    geo.idx_events = [None] * len(geo.idx_events)
    geo.nn_events = {i: None for i in geo.nn_events}
    return geo


def voxelize_batch(points, dataset, max_points=5, max_voxels=40000):
    """(F, P, 4) -> the dict MeanVFE / the sparse trunk consume (voxels, counts, coords [b, z, y, x]): the reference's host
    voxeliser call (pcdet/datasets/processor/data_processor.py:15-60, 132-152) for all clouds of the batch at once, on the
    tensors' device, with the same semantics (first-appearance voxel order, first max_points points, max_voxels cap)."""
    from .pcdet.datasets.processor.data_processor import points_to_voxels_batch
    return points_to_voxels_batch(points, dataset.point_cloud_range, dataset.voxel_size, max_points, max_voxels)


def voxel_route_capturable():
    """Whether the voxel route can be recorded into a HIP graph: its voxeliser has data-dependent sizes (torch.unique ->
    host sync), so not yet."""
    from .pcdet.datasets.processor import data_processor
    return bool(getattr(data_processor, "FIXED_CAPACITY", False))


def synthetic_loss(outputs):
    """Scalar over all 16 outputs so that every trainable parameter receives a gradient (round-1 placeholder objective;
    kept for tests that need a label-free scalar)."""
    return sum((o.float() ** 2).mean() for o in outputs)


def reference_loss(outputs, batch):
    """The reference's training objective (train_func.py:166-256, Loss = "L_total" as train_func.py:553 selects it: weighted
    adjacency BCE + individual pose CE / interaction BCE + social-group pose / interaction BCE) on synthetic annotations,
    evaluated on the device without a Python loop over scenes (losses.mgar_losses_uniform).  Every frame-scene is one sample;
    the per-sample terms are summed over the scenes, which is what the reference accumulates at its BATCH_SIZE 1
    (train_func.py:260-269).  As in the reference, the cardinality head is not part of L_total and receives no gradient."""
    from . import losses
    return losses.mgar_losses_uniform(outputs, batch["social_group_id"], batch["action"], batch["social_group_activity"],
                                      batch["n_actors"], Loss="L_total", reference_semantics=False)["L_total"]


class TrainStep:
    """model + Adam + data parallelism; ``run(batch)`` = forward, loss, backward, (gradient all-reduce,) optimizer step.

    ddp=True wraps the model in DistributedDataParallel (bucketed all-reduce overlapped with backward).
    ``capture(batch)`` records forward + backward of one step into a HIP graph (torch.cuda.CUDAGraph): the ~1 900-
    4 700 launches of a step are replayed without the host, which is what bounds a rank that holds a single clip.
    In graph mode the data-parallel exchange is ONE all-reduce of the flattened gradients after the replay
    (no DDP wrapper: its hooks cannot live inside a captured backward), then Adam."""

    def __init__(self, n_actors, n_points, device, gat=True, route="pointnet2", ddp=False, lr=1e-3, seed=2023,
                 manual_allreduce=False, loss="reference"):
        assert loss in ("reference", "synthetic")
        self.loss_kind = loss
        torch.manual_seed(seed)  # the reference seeds 2023 (train_func.py:45-47)
        self.model = ClipModel(n_actors, n_points, gat, route).to(device)
        self.model.train()
        self.module = self.model
        self.manual_allreduce = bool(manual_allreduce)
        # The reference objective is a SUM over the scenes of the batch (what train_func.py:260-269 accumulates over its
        # micro-steps), so the gradient of the GLOBAL batch is the SUM of the ranks' gradients: all-reduce(SUM), no division --
        # an N-rank step is then the same update as the one-rank step on the same global batch.  The label-free
        # synthetic objective is a mean, whose global gradient is the ranks' average.
        self.grad_average = (loss == "synthetic")
        self._ddp_wrapped = bool(ddp and not manual_allreduce)
        if ddp and not manual_allreduce:
            from torch.nn.parallel import DistributedDataParallel as DDP
            dev_ids = [device.index] if device.type == "cuda" else None
            # the reference objective leaves the cardinality head (and, on the voxel route, conv_out / shared_fc) without a
            # gradient on every rank: the reducer has to be told, or the buckets holding them are never exchanged
            self.model = DDP(self.model, device_ids=dev_ids, find_unused_parameters=(loss == "reference"),
                             gradient_as_bucket_view=True, bucket_cap_mb=64)
        if ddp and manual_allreduce:
            self._sync_from_rank0()
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        # train_func.py:552 Adam(lr=1e-3); on the device the fused implementation (a handful of launches instead of ~30)
        self.opt = torch.optim.Adam(self.params, lr=lr, fused=(device.type == "cuda") or None)
        self.graph = None
        self._static_batch = None
        self._loss = None

    # ---- eager step ------------------------------------------------------------------------------
    def _forward_backward(self, batch):
        self.opt.zero_grad(set_to_none=True)
        out = self.model(batch)
        loss = self._loss_of(out, batch)
        if self._ddp_wrapped and not self.grad_average:
            # DistributedDataParallel averages the ranks' gradients; a summed objective wants their sum
            import torch.distributed as dist
            (loss * dist.get_world_size()).backward()
        else:
            loss.backward()
        self.module.finish_prefetch()
        return loss.detach()

    def _loss_of(self, out, batch):
        return reference_loss(out, batch) if self.loss_kind == "reference" else synthetic_loss(out)

    def _sync_from_rank0(self):
        """Flat-all-reduce data parallelism: every rank starts from rank 0's parameters and buffers (DistributedDataParallel
        does this itself).  BatchNorm running statistics then evolve per rank -- they are never exchanged, as under the
        reference's nn.DataParallel, whose replicas' buffer updates are discarded except replica 0's (train_func.py:512)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t, src=0)

    def _exchange_gradients(self):
        """One all-reduce(SUM) over all gradients, flattened (~119 MB fp32 per step); divided by the
        world size only for an averaged objective (see ``grad_average``)."""
        if not self.manual_allreduce:
            return
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        # parameters outside the objective (the cardinality head under L_total; conv_out / shared_fc of the voxel route, unused
        # downstream as in the reference) have no gradient on ANY rank: the layout below is the same everywhere.  A parameter
        # that received a gradient on some ranks only would desynchronise the collective, so the set is pinned at the first step.
        have = tuple(p.grad is not None for p in self.params)
        if getattr(self, "_grad_layout", None) is None:
            self._grad_layout = have
        assert have == self._grad_layout, "the set of parameters with gradients changed between steps"
        grads = [p.grad for p in self.params if p.grad is not None]
        flat = torch._utils._flatten_dense_tensors(grads)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if self.grad_average:
            flat.div_(dist.get_world_size())
        for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(f)

    def run_eager(self, batch):
        loss = self._forward_backward(batch)
        self._exchange_gradients()
        self.opt.step()
        return loss

    # ---- HIP-graph step ---------------------------------------------------------------------------
    def capture(self, batch, warmup=2):
        """Record forward + backward on ``batch`` (its tensors become the static inputs: later batches are copied
        into them).  `warmup` eager steps run first on a side stream, as graph capture requires."""
        assert self.graph is None and batch["images"].is_cuda
        assert not hasattr(self.model, "module") or self.model is self.module, "capture() needs manual_allreduce, not DDP"
        self._static_batch = batch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.run_eager(batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self.model(batch)
            loss = self._loss_of(out, batch)
            loss.backward()
            self.module.finish_prefetch()
        self.graph, self._loss = graph, loss.detach()
        self._graph_grads = [p.grad for p in self.params]   # the buffers every replay writes (graph-private pool)
        return self

    def run(self, batch):
        if self.graph is None:
            return self.run_eager(batch)
        if batch is not self._static_batch:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self._static_batch[k].copy_(v)
        self.graph.replay()
        for p, g in zip(self.params, self._graph_grads):      # an eager step in between re-points .grad elsewhere
            p.grad = g
        self._exchange_gradients()
        self.opt.step()
        return self._loss


class ForwardStep:
    """Train-mode forward of the clip model (batch-statistics BatchNorm, dropout active) without autograd: the workload of
    BASELINE configs c2 / c5.  precision="bf16": feature payloads and GEMMs / convolutions in bf16 -- the hand-written
    kernels take bf16 payloads directly (include/mgar_ops.h, `_bf16` entry points: fp32 arithmetic inside, fp32 BatchNorm
    statistics), library GEMMs / convolutions run under torch.autocast(bfloat16) -- while coordinates, distances, indices
    and the per-scene fusion net stay fp32 / int32 (SURVEY.md section 8 header), so every index tensor is bit-identical to
    the fp32 run.  Same interface as TrainStep (capture / run / run_eager, .module, .graph)."""

    def __init__(self, n_actors, n_points, device, gat=True, route="pointnet2", precision="fp32", seed=2023):
        assert precision in ("fp32", "bf16")
        torch.manual_seed(seed)
        self.model = ClipModel(n_actors, n_points, gat, route).to(device)
        self.model.train()
        self.module = self.model
        self.precision = precision
        self.device = device
        if precision == "bf16":
            # the frozen I3D's convolution weights are converted once (autocast would re-cast them every step);
            # BatchNorm vectors and every trainable parameter stay fp32
            for m in self.model.net.RGB_backbone.backbone_net.modules():
                if isinstance(m, nn.Conv3d):
                    m.weight.data = m.weight.data.to(torch.bfloat16)
        self.graph = None
        self._static_batch = None
        self._out = None

    def _forward(self, batch):
        with torch.no_grad(), torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=self.precision == "bf16"):
            return self.model(batch)

    def run_eager(self, batch):
        return self._forward(batch)

    def capture(self, batch, warmup=2):
        assert self.graph is None and batch["images"].is_cuda
        self._static_batch = batch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._forward(batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = self._forward(batch)
        self.graph, self._out = graph, out
        return self

    def run(self, batch):
        if self.graph is None:
            return self._forward(batch)
        if batch is not self._static_batch:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self._static_batch[k].copy_(v)
        self.graph.replay()
        return self._out


def trainable_parameter_count(module):
    return sum(p.numel() for p in module.parameters() if p.requires_grad)
