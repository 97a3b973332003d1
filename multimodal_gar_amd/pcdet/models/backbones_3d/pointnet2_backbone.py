"""Scene-level PointNet++ (MSG) encoder/decoder: SA x k then FP x k on dense batches.
Mirror of the reference's pcdet/models/backbones_3d/pointnet2_backbone.py:9-95 (PointNet2MSG).
The stacked variant (PointNet2Backbone, :97-206) asserts False in its constructor in the
reference and is not provided."""
import torch
import torch.nn as nn

from ...ops.pointnet2.pointnet2_batch import pointnet2_modules


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
        # optional: centres picked ahead of time by the caller (same FPS, issued earlier so that it overlaps other
        # work -- one workgroup per cloud leaves most of the chip idle); SA modules take new_xyz as in the reference
        pre = batch_dict.get('sa_new_xyz') or []
        for k, sa in enumerate(self.SA_modules):
            li_xyz, li_features = sa(l_xyz[-1], l_features[-1], new_xyz=pre[k] if k < len(pre) else None)
            l_xyz.append(li_xyz)
            l_features.append(li_features)
        for i in range(-1, -(len(self.FP_modules) + 1), -1):
            l_features[i - 1] = self.FP_modules[i](l_xyz[i - 1], l_xyz[i], l_features[i - 1], l_features[i])
        # (B, C, n) channel-major, as produced: consumers on the device take this (no transposed copy of ~1 GB)
        batch_dict['point_features_cm'] = l_features[0]
        if self.model_cfg.get('STACKED_POINT_FEATURES', True) or not xyz.is_cuda:   # the reference's key (pointnet2_backbone.py:91-92)
            point_features = l_features[0].permute(0, 2, 1).contiguous()
            batch_dict['point_features'] = point_features.view(-1, point_features.shape[-1])
        batch_dict['point_coords'] = torch.cat((batch_idx[:, None].float(), l_xyz[0].view(-1, 3)), dim=1)
        return batch_dict
