"""Scene-level PointNet++ (MSG) encoder/decoder: SA x k then FP x k on dense batches.
Mirror of the reference's pcdet/models/backbones_3d/pointnet2_backbone.py:9-95 (PointNet2MSG).
The stacked variant (PointNet2Backbone, :97-206) asserts False in its constructor in the
reference and is not provided."""
import torch
import torch.nn as nn

from ...ops.pointnet2.pointnet2_batch import pointnet2_modules


class TrunkGeometry:
    """Results of PointNet2MSG.geometry() with the events that order them against the consumer's stream."""

    def __init__(self):
        self.centres, self.ball_idx, self.centre_events, self.idx_events = [], [], [], []
        self.nn, self.nn_events = {}, {}

    @staticmethod
    def _mark():
        return torch.cuda.current_stream().record_event() if torch.cuda.is_available() else None

    @staticmethod
    def _wait(ev):
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def add_level(self, centres, idx):
        self.centres.append(centres)
        self.ball_idx.append(idx)
        self.centre_events.append(self._mark())
        self.idx_events.append(None)

    def set_ball_idx(self, k, idx):
        self.ball_idx[k] = idx
        self.idx_events[k] = self._mark()

    def add_neighbours(self, i, idx_weight):
        self.nn[i] = idx_weight
        self.nn_events[i] = self._mark()

    def level(self, k):
        self._wait(self.idx_events[k] if self.idx_events[k] is not None else self.centre_events[k])
        return self.centres[k], self.ball_idx[k]

    def neighbours(self, i):
        self._wait(self.nn_events[i])
        return self.nn[i]


class PointNet2MSG(nn.Module):
    def __init__(self, model_cfg, input_channels, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        sa = model_cfg.SA_CONFIG
        self.SA_modules = nn.ModuleList()
        channel_in = input_channels - 3
        skip_channels = [channel_in]
        channel_out = channel_in
        for k in range(len(sa.NPOINTS)):
            mlps = [[channel_in] + list(m) for m in sa.MLPS[k]]
            channel_out = sum(m[-1] for m in mlps)
            self.SA_modules.append(pointnet2_modules.PointnetSAModuleMSG(
                npoint=sa.NPOINTS[k], radii=sa.RADIUS[k], nsamples=sa.NSAMPLE[k], mlps=mlps,
                use_xyz=sa.get('USE_XYZ', True)))
            skip_channels.append(channel_out)
            channel_in = channel_out
        self.FP_modules = nn.ModuleList()
        fp = model_cfg.FP_MLPS
        for k in range(len(fp)):
            pre = fp[k + 1][-1] if k + 1 < len(fp) else channel_out
            self.FP_modules.append(pointnet2_modules.PointnetFPModule(mlp=[pre + skip_channels[k]] + list(fp[k])))
        self.num_point_features = fp[0][-1]

    def geometry(self, points, stream=None, levels=None, balls=True, neighbours=True):
        """Everything of the trunk that depends on coordinates only -- per SA level the FPS centres and the ball queries of
        the folded scales, per FP level the 3-NN interpolation weights -- issued on ``stream`` (default: the current one)
        ahead of the feature path.  points (B, n, >= 3).  -> TrunkGeometry for batch_dict['trunk_geometry']; same values as
        the modules compute.  Everything, the slicing of ``points`` included, is issued on ``stream``.  ``levels`` / ``balls`` /
        ``neighbours`` restrict it (first SA levels only / no ball queries / no 3-NN); the modules compute the rest inline."""
        import contextlib
        geo = TrunkGeometry()
        with torch.no_grad(), (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            l_xyz = [points[..., :3].contiguous()]
            n_feat = self.SA_modules[0].mlps[0][0].in_channels - 3
            for sa in self.SA_modules[:levels]:
                centres = sa.pick_centres(l_xyz[-1])
                geo.add_level(centres, None)                     # the centres first: the feature path of level 1 waits for them
                idx = sa.ball_indices(l_xyz[-1], centres, n_feat if n_feat > 0 else None) if balls else None
                geo.set_ball_idx(len(l_xyz) - 1, idx)
                l_xyz.append(centres)
                n_feat = sum(m[-3].out_channels for m in sa.mlps)
            for i in range(-1, -(len(self.FP_modules) + 1), -1):
                if neighbours and levels is None:
                    geo.add_neighbours(i, self.FP_modules[i].neighbour_weights(l_xyz[i - 1], l_xyz[i]))
        return geo

    @staticmethod
    def break_up_pc(pc):
        return pc[:, 0], pc[:, 1:4].contiguous(), (pc[:, 4:].contiguous() if pc.size(-1) > 4 else None)

    def forward(self, batch_dict):
        """points (num_points, 4 + C) [batch_idx, x, y, z, ...] with equal counts per sample ->
        point_features (N, C), point_coords (N, 4)."""
        batch_size = batch_dict['batch_size']
        batch_idx, xyz, features = self.break_up_pc(batch_dict['points'])
        xyz = xyz.view(batch_size, -1, 3)
        if features is not None:
            features = features.view(batch_size, -1, features.shape[-1]).permute(0, 2, 1).contiguous()
            if features.is_cuda and torch.is_autocast_enabled():
                # bf16 configurations: the per-point payload is bf16 from the start, so the grouped tensors of level 1 are
                # written in bf16 by the fused query-and-group kernel (coordinates stay fp32)
                features = features.to(torch.get_autocast_dtype('cuda'))
        l_xyz, l_features = [xyz], [features]
        # optional: the trunk's GEOMETRY (centres, ball queries, 3-NN weights depend on coordinates only) computed ahead of
        # time by the caller on another stream -- same values; SA / FP modules take them as the reference takes new_xyz
        geo = batch_dict.get('trunk_geometry')
        pre = batch_dict.get('sa_new_xyz') or []
        for k, sa in enumerate(self.SA_modules):
            if geo is not None and k < len(geo.centres):
                centres, ball_idx = geo.level(k)
                li_xyz, li_features = sa(l_xyz[-1], l_features[-1], new_xyz=centres, pre_idx=ball_idx)
            else:
                li_xyz, li_features = sa(l_xyz[-1], l_features[-1], new_xyz=pre[k] if k < len(pre) else None)
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i],
                                                   nn_weights=geo.neighbours(i) if geo is not None and i in geo.nn else None)
        # (B, C, n) channel-major, as produced: consumers on the device take this (no transposed copy of ~1 GB)
        batch_dict['point_features_cm'] = l_features[0]
        if self.model_cfg.get('STACKED_POINT_FEATURES', True) or not xyz.is_cuda:   # the reference's key (pointnet2_backbone.py:91-92)
            point_features = l_features[0].permute(0, 2, 1).contiguous()
            batch_dict['point_features'] = point_features.view(-1, point_features.shape[-1])
        batch_dict['point_coords'] = torch.cat((batch_idx[:, None].float(), l_xyz[0].view(-1, 3)), dim=1)
        return batch_dict
