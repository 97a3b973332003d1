"""Set-abstraction / feature-propagation modules on stacked batches.

Mirror of the reference's pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:1-157
(build_local_aggregation_module, StackSAModuleMSG, StackPointnetFPModule).  The
VectorPool* modules (:160-470) are out of scope.
"""
from typing import List

import torch
import torch.nn as nn

from . import pointnet2_utils
from .....nn_utils import PointwiseSequential


def _shared_mlp_2d(spec):
    layers = []
    for c_in, c_out in zip(spec[:-1], spec[1:]):
        layers += [nn.Conv2d(c_in, c_out, kernel_size=1, bias=False), nn.BatchNorm2d(c_out), nn.ReLU()]
    return PointwiseSequential(*layers)


def build_local_aggregation_module(input_channels, config):
    name = config.get('NAME', 'StackSAModuleMSG')
    if name != 'StackSAModuleMSG':
        raise NotImplementedError('%s (vector-pool aggregation is out of scope)' % name)
    mlps = config.MLPS
    for k in range(len(mlps)):
        mlps[k] = [input_channels] + mlps[k]
    layer = StackSAModuleMSG(radii=config.POOL_RADIUS, nsamples=config.NSAMPLE, mlps=mlps, use_xyz=True,
                             pool_method='max_pool')
    return layer, sum(x[-1] for x in mlps)


class StackSAModuleMSG(nn.Module):
    """Reference: pointnet2_modules.py:30-112 (kaiming-normal convs, BN weight 1 / bias 0)."""

    # the folded path's grouping backward without float atomics (csrc/query_group.hip, qg_stack_bwd_rows_kernel); False: the
    # atomic scatter of round 1 (kept for the A/B in profiles/ and for C_k > 64)
    rowmajor_grad = True

    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(_shared_mlp_2d(spec))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def _rows_bwd(self, k):
        """Scale k's grouping backward on the atomic-free rows path: its MLP must go on after the first BatchNorm + ReLU (else
        that BatchNorm is fused with the max-pool and its backward is channel-major) with at most 64 channels."""
        mlp = self.mlps[k]
        return bool(self.rowmajor_grad) and len(mlp) > 3 and mlp[0].out_channels <= 64 and torch.is_grad_enabled()

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        """-> (new_xyz (M, 3), new_features (M, sum_k mlps[k][-1])).

        `features` may also be (B, C, n) channel-major with n points in every sample (device path only).  The
        returned new_features has the reference's shape; on the device its memory is channel-major (a transposed
        view), which is what both its producer (the fused BN/ReLU/max kernel) and its consumer (the (N, C, 6, 6, 6)
        grid of the non-local block) want -- two strided transposing copies become contiguous ones."""
        per_scale = []
        projected = {}
        n_feat = None if features is None else features.shape[1]
        if features is not None and features.dim() == 3:
            assert self.pool_method == 'max_pool' and xyz.is_cuda, "channel-major features: fused device path only"
        if self.pool_method == 'max_pool' and features is not None and xyz.is_cuda:
            # "project, then group": layer 0 is linear, apply its feature half to the N points first --
            # for all scales in ONE GEMM over the shared features
            fold = [k for k, (g, mlp) in enumerate(zip(self.groupers, self.mlps))
                    if g.use_xyz and mlp.first_layer_foldable(3 + n_feat)]
            if fold:
                ys = pointnet2_utils._FusedQueryGroupProjMSG.apply(
                    xyz, xyz_batch_cnt.int(), new_xyz, new_xyz_batch_cnt.int(), features,
                    tuple(self.groupers[k].radius for k in fold), tuple(self.groupers[k].nsample for k in fold),
                    tuple(self._rows_bwd(k) for k in fold), *[self.mlps[k][0].weight for k in fold])
                projected = dict(zip(fold, ys[:len(fold)]))
                proj_stats = dict(zip(fold, ys[len(fold):]))
        for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            if k in projected:
                y0 = projected[k]
                st = proj_stats[k] if proj_stats[k].numel() else None
                x = mlp.forward_maxpool(y0.view(1, y0.shape[0], new_xyz.shape[0], -1), start=1, rowmajor_input_grad=self._rows_bwd(k),
                                        input_stats=st)
                per_scale.append(x.squeeze(0).permute(1, 0))
                continue
            assert features is None or features.dim() == 2, "channel-major features need the projected (foldable) path"
            grouped, _ = grouper.forward_channel_major(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)
            grouped = grouped.view(1, grouped.shape[0], new_xyz.shape[0], -1)              # (1, C, M, nsample)
            if self.pool_method == 'max_pool':
                x = mlp.forward_maxpool(grouped)                                            # (1, C', M)
            elif self.pool_method == 'avg_pool':
                x = mlp(grouped).mean(dim=3)
            else:
                raise NotImplementedError
            per_scale.append(x.squeeze(0).permute(1, 0))                                    # (M, C')
        if xyz.is_cuda and all(p.stride(0) == 1 for p in per_scale):
            return new_xyz, torch.cat([p.t() for p in per_scale], dim=0).t()                # (M, C'), channel-major memory
        return new_xyz, torch.cat(per_scale, dim=1)


class StackPointnetFPModule(nn.Module):
    """Reference: pointnet2_modules.py:115-157."""

    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = _shared_mlp_2d(mlp)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        inv = 1.0 / (dist + 1e-8)
        weight = inv / torch.sum(inv, dim=-1, keepdim=True)
        spread = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        merged = spread if unknown_feats is None else torch.cat([spread, unknown_feats], dim=1)
        x = self.mlp(merged.permute(1, 0)[None, :, :, None])           # (1, C, N, 1)
        return x.squeeze(0).squeeze(-1).permute(1, 0)
