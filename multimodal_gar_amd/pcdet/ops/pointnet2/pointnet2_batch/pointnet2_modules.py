"""Set-abstraction / feature-propagation modules on dense batches.

Mirror of the reference's pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py
(_PointnetSAModuleBase, PointnetSAModuleMSG, PointnetSAModule, PointnetFPModule): same
constructor keywords, sub-module names (``groupers``, ``mlps``, ``mlp``) and parameter
shapes, so reference state dicts load.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import pointnet2_utils
from .....nn_utils import PointwiseSequential


def shared_mlp_2d(spec: List[int]) -> nn.Sequential:
    """[Conv2d 1x1 (no bias) -> BatchNorm2d -> ReLU] per consecutive channel pair."""
    layers = []
    for c_in, c_out in zip(spec[:-1], spec[1:]):
        layers += [nn.Conv2d(c_in, c_out, kernel_size=1, bias=False), nn.BatchNorm2d(c_out), nn.ReLU()]
    return PointwiseSequential(*layers)


def pool_over_samples(x: torch.Tensor, method: str) -> torch.Tensor:
    """(B, C, npoint, nsample) -> (B, C, npoint)."""
    if method == 'max_pool':
        return x.max(dim=3).values
    if method == 'avg_pool':
        return x.mean(dim=3)
    raise NotImplementedError(method)


class _PointnetSAModuleBase(nn.Module):
    rowmajor_grad = True    # folded scales: grouping backward without LDS / global float atomics (False: round-2 path, A/B)

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None
        self.pool_method = 'max_pool'

    def pick_centres(self, xyz):
        """Farthest point sampling + gather: the centres of this level (reference pointnet2_modules.py:33-41)."""
        picked = pointnet2_utils.farthest_point_sample(xyz, self.npoint)
        xyz_t = xyz.transpose(1, 2).contiguous()
        return pointnet2_utils.gather_operation(xyz_t, picked).transpose(1, 2).contiguous()

    def _fold_scales(self, n_feat, on_device):
        """Scales whose first layer is applied to the points before the grouping ("project, then group"): device only."""
        if not (self.pool_method == 'max_pool' and n_feat is not None and on_device):
            return []
        return [k for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps))
                if isinstance(grouper, pointnet2_utils.QueryAndGroup) and grouper.use_xyz and mlp.first_layer_foldable(3 + n_feat)]

    def ball_indices(self, xyz, new_xyz, n_feat):
        """The ball queries of all scales in ONE scan of the cloud (they share centres and cloud) -> {scale: idx}; device only.
        Pure geometry: callable ahead of the features (model/../workload.py can issue it on a side stream)."""
        del n_feat
        scales = [k for k, g in enumerate(self.groupers) if isinstance(g, pointnet2_utils.QueryAndGroup)] if xyz.is_cuda else []
        if len(scales) > 1 and hasattr(pointnet2_utils.pointnet2, "ball_query_multi_wrapper"):
            return dict(zip(scales, pointnet2_utils.ball_query_multi([self.groupers[k].radius for k in scales],
                                                                     [self.groupers[k].nsample for k in scales], xyz, new_xyz)))
        return {}

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor] = None,
                new_xyz=None, pre_idx=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B, N, 3), features (B, C, N) -> new_xyz (B, npoint, 3), (B, sum_k mlps[k][-1], npoint).
        Reference: pointnet2_modules.py:19-55.  ``new_xyz`` / ``pre_idx``: this level's centres / ball_indices() computed
        ahead of time by the caller (same values)."""
        if new_xyz is None and self.npoint is not None:
            new_xyz = self.pick_centres(xyz)
        per_scale = []
        n_feat = None if features is None else features.shape[1]
        fold = self._fold_scales(n_feat, xyz.is_cuda)
        if pre_idx is None:
            pre_idx = self.ball_indices(xyz, new_xyz, n_feat)
        for k, (grouper, mlp) in enumerate(zip(self.groupers, self.mlps)):
            if k in fold:
                # "project, then group": layer 0 is linear, apply its feature half to the N points first.  rows: the grouping
                # backward on the atomic-free rows kernels (needs the MLP to go on after its first BatchNorm + ReLU -- else that
                # BatchNorm is fused with the max-pool and its backward is channel-major -- with at most 64 channels)
                rows = self.rowmajor_grad and len(mlp) > 3 and mlp[0].out_channels <= 64 and torch.is_grad_enabled()
                y0 = grouper.forward_projected(xyz, new_xyz, features, mlp[0].weight, idx=pre_idx.get(k), rows_bwd=rows)
                per_scale.append(mlp.forward_maxpool(y0, start=1, rowmajor_input_grad=rows))
                continue
            grouped = grouper(xyz, new_xyz, features, idx=pre_idx[k]) if k in pre_idx else grouper(xyz, new_xyz, features)   # (B, C', npoint, nsample)
            if self.pool_method == 'max_pool':
                per_scale.append(mlp.forward_maxpool(grouped))             # BN + ReLU + max fused on the device
            else:
                per_scale.append(pool_over_samples(mlp(grouped), self.pool_method))
        return new_xyz, torch.cat(per_scale, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping (reference pointnet2_modules.py:58-100).
    As in the reference, ``mlps[i][0]`` is incremented IN PLACE by 3 when use_xyz."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]],
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(shared_mlp_2d(spec))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction (reference pointnet2_modules.py:103-119)."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None,
                 bn: bool = True, use_xyz: bool = True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method)


class PointnetFPModule(nn.Module):
    """Feature propagation: 3-NN inverse-distance interpolation + skip concat + shared MLP
    (reference pointnet2_modules.py:122-170)."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = shared_mlp_2d(mlp)

    project_first = True    # device path: "project, then interpolate" (False: interpolate, concatenate, convolve -- the reference's order)

    def _project_then_interpolate(self, idx, weight, unknow_feats, known_feats):
        """The module's forward with its first layer applied BEFORE the interpolation (round 3).  The first shared-MLP layer is
        a bias-free 1x1 convolution W = [W_a | W_b] over cat([interp(f), skip]) and the three-point interpolation is linear per
        channel, so  W cat([interp(f), skip]) = interp(W_a f) + W_b skip:  W_a f is a GEMM over the m KNOWN points (a quarter of
        the n unknown ones at every level of PointNet2MSG), the interpolation moves C_1 <= C_known channels, and the (C_known + C_skip,
        n) concatenated tensor is never built (2 GB per step at the finest level of config c3).  Same sums in another order.
        None where it does not apply (CPU, no skip features, a first layer with a bias or wider than the known features)."""
        from .....nn_utils import _is_pointwise
        if not (self.project_first and known_feats.is_cuda and unknow_feats is not None and len(self.mlp) and _is_pointwise(self.mlp[0])
                and self.mlp[0].bias is None and hasattr(pointnet2_utils.pointnet2, "three_interpolate_add_wrapper")):
            return None
        conv = self.mlp[0]
        c_known, c_skip = known_feats.shape[1], unknow_feats.shape[1]
        if conv.in_channels != c_known + c_skip or conv.out_channels > c_known or known_feats.dtype != unknow_feats.dtype:
            return None
        w = conv.weight.view(conv.out_channels, conv.in_channels).to(known_feats.dtype)
        b = known_feats.shape[0]
        zf = torch.bmm(w[:, :c_known].unsqueeze(0).expand(b, -1, -1), known_feats)                 # (B, C_1, m)
        zs = torch.bmm(w[:, c_known:].unsqueeze(0).expand(b, -1, -1), unknow_feats)               # (B, C_1, n)
        z1 = pointnet2_utils.ThreeInterpolateAdd.apply(zf.contiguous(), idx, weight, zs.contiguous())
        return self.mlp._run(z1.unsqueeze(-1), False, start=1)[0].squeeze(-1)

    @staticmethod
    def neighbour_weights(unknown, known):
        """three_nn + inverse-distance weights (reference pointnet2_modules.py:137-140) -> (idx, weight): pure geometry."""
        dist, idx = pointnet2_utils.three_nn(unknown, known)
        inv = 1.0 / (dist + 1e-8)
        return idx, inv / torch.sum(inv, dim=2, keepdim=True)

    def forward(self, unknown: torch.Tensor, known: torch.Tensor, unknow_feats: torch.Tensor,
                known_feats: torch.Tensor, nn_weights=None) -> torch.Tensor:
        """``nn_weights``: neighbour_weights(unknown, known) computed ahead of time by the caller (same values)."""
        if known is not None:
            idx, weight = nn_weights if nn_weights is not None else self.neighbour_weights(unknown, known)
            y = self._project_then_interpolate(idx, weight, unknow_feats, known_feats)
            if y is not None:
                return y
            if unknow_feats is not None and known_feats.is_cuda:
                # interpolation written straight into the concatenated tensor (no copy of the interpolated half)
                merged = pointnet2_utils.three_interpolate_concat(known_feats, idx, weight, unknow_feats)
                return self.mlp(merged.unsqueeze(-1)).squeeze(-1)
            spread = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            spread = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        merged = spread if unknow_feats is None else torch.cat([spread, unknow_feats], dim=1)
        return self.mlp(merged.unsqueeze(-1)).squeeze(-1)
