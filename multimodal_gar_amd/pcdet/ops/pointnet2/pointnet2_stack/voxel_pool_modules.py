"""Voxel RoI pooling layer.

Mirror of the reference's pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py
(NeighborVoxelSAModuleMSG): same constructor keywords and sub-module names (groupers,
mlps_in, mlps_pos, mlps_out).
"""
from typing import List

import torch
import torch.nn as nn

from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import pointnet2_stack_cuda as pointnet2
from . import voxel_query_utils
from .....nn_utils import PointwiseSequential


class _FusedVoxelRoIPool(Function):
    """pooled (C, M) = max_s relu(feats[idx[m, s]] + BN_pos(w_pos . rel_xyz)) on csrc/voxel_roi_pool.hip: everything
    between mlps_in and mlps_out of one scale (reference voxel_pool_modules.py:94-117) without any (C, M, nsample) tensor."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats, idx_raw, w_pos, gamma, beta, bn):
        n_query, nsample = idx_raw.shape
        chans = feats.shape[1]
        xyz, new_xyz, feats = xyz.contiguous(), new_xyz.contiguous(), feats.contiguous()
        w_pos, gamma, beta = w_pos.contiguous().float(), gamma.float(), beta.float()
        dev = xyz.device
        train_stats = bn.training or not bn.track_running_stats
        moments = None
        if train_stats:
            mean = torch.empty((chans,), dtype=torch.float32, device=dev)
            invstd = torch.empty_like(mean)
            moments = torch.empty((10,), dtype=torch.float64, device=dev)
            track = bn.track_running_stats and bn.running_mean is not None
            pointnet2.voxel_roi_pool_stats(n_query, nsample, chans, xyz, new_xyz, idx_raw, w_pos, bn.eps,
                                           bn.momentum if bn.momentum is not None else 0.1, moments, mean, invstd,
                                           bn.running_mean if track else None, bn.running_var if track else None,
                                           bn.num_batches_tracked if track else None)
        else:
            mean, invstd = bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)
        pooled = torch.empty((chans, n_query), dtype=feats.dtype, device=dev)
        arg = torch.empty((chans, n_query), dtype=torch.uint8, device=dev)
        pointnet2.voxel_roi_pool_fwd(n_query, nsample, chans, xyz, new_xyz, feats, idx_raw, w_pos, mean, invstd, gamma, beta, pooled, arg)
        ctx.save_for_backward(xyz, new_xyz, idx_raw, w_pos, mean, invstd, gamma, moments, pooled, arg)
        ctx.meta = (feats.shape, train_stats)
        ctx.mark_non_differentiable(arg)
        return pooled, arg

    @staticmethod
    @once_differentiable
    def backward(ctx, dpooled, darg=None):
        xyz, new_xyz, idx_raw, w_pos, mean, invstd, gamma, moments, pooled, arg = ctx.saved_tensors
        (n_rows, chans), train_stats = ctx.meta
        n_query, nsample = idx_raw.shape
        dfeats = torch.zeros((n_rows, chans), dtype=torch.float32, device=xyz.device) if ctx.needs_input_grad[2] else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        dw = torch.empty_like(w_pos)
        pointnet2.voxel_roi_pool_bwd(n_query, nsample, chans, xyz, new_xyz, idx_raw, w_pos, mean, invstd, gamma, moments, train_stats,
                                     dpooled.contiguous().float(), pooled.float(), arg, dfeats, dgamma, dbeta, dw)
        return None, None, dfeats, None, dw, dgamma, dbeta, None


class _RowsLinear(Function):
    """z = x W^T for row-major x (N, C_in) with N in the millions and W (C_out, C_in) tiny.  Forward and dX are plain GEMMs; the
    weight gradient dz^T x -- a (C_out x C_in) result reduced over N rows, which the library runs at 2-3 TFLOP/s (4.3 ms at
    N = 4.15 M, C 32) -- goes to csrc/rowmajor_dw.hip (both operands streamed once through the MFMA)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        dx = dz @ w if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = pointnet2.rowmajor_dw(dz, x) if (x.is_cuda and w.shape[0] <= 96 and w.shape[1] <= 128) else dz.t() @ x
        return dx, dw


class NeighborVoxelSAModuleMSG(nn.Module):
    def __init__(self, *, query_ranges: List[List[int]], radii: List[float],
                 nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(query_ranges) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps_in = nn.ModuleList()
        self.mlps_pos = nn.ModuleList()
        self.mlps_out = nn.ModuleList()
        for max_range, nsample, radius, spec in zip(query_ranges, nsamples, radii, mlps):
            self.groupers.append(voxel_query_utils.VoxelQueryAndGrouping(max_range, radius, nsample))
            self.mlps_in.append(PointwiseSequential(nn.Conv1d(spec[0], spec[1], kernel_size=1, bias=False),
                                              nn.BatchNorm1d(spec[1])))
            self.mlps_pos.append(PointwiseSequential(nn.Conv2d(3, spec[1], kernel_size=1, bias=False),
                                               nn.BatchNorm2d(spec[1])))
            self.mlps_out.append(PointwiseSequential(nn.Conv1d(spec[1], spec[2], kernel_size=1, bias=False),
                                               nn.BatchNorm1d(spec[2]), nn.ReLU()))
        self.relu = nn.ReLU()
        self.pool_method = pool_method
        self.fused = True      # device path: csrc/voxel_roi_pool.hip (set False to run the un-fused op chain on the device)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def _mlp_in_rows(self, k, features):
        """mlps_in[k] = Conv1d(1x1, no bias) + BatchNorm1d on ALL voxels (reference voxel_pool_modules.py:86-88), evaluated on
        the row-major (N, C) features as they are: one GEMM + the row-major BatchNorm kernels (csrc/channels_last.hpp), without
        the two transposing copies of the (1, C, N) formulation (7.6 + 6 ms per step at config c3).  Device + train mode only."""
        conv, bn = self.mlps_in[k][0], self.mlps_in[k][1]
        if not (features.is_cuda and features.dtype == torch.float32 and bn.training and conv.bias is None and len(self.mlps_in[k]) == 2):
            return None
        from .....bn_ops import bn_act_rows
        z = _RowsLinear.apply(features.contiguous(), conv.weight.view(conv.out_channels, conv.in_channels))   # (N, C)
        return bn_act_rows(z, bn, False)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords, features, voxel2point_indices):
        """xyz (N, 3) voxel centres, features (N, C_in), new_xyz (M, 3) grid points,
        new_coords (M, 4) [b, x, y, z] -> (M, sum_k mlps[k][-1]).
        Reference: voxel_pool_modules.py:70-130."""
        new_coords = new_coords[:, [0, 3, 2, 1]].contiguous()  # -> [b, z, y, x]
        per_scale = []
        for k, grouper in enumerate(self.groupers):
            feats_in = self._mlp_in_rows(k, features)
            if feats_in is None:
                feats_in = self.mlps_in[k](features.permute(1, 0).unsqueeze(0))    # (1, C, N)
                feats_in = feats_in.squeeze(0).permute(1, 0).contiguous()          # (N, C)
            pos_conv, pos_bn = self.mlps_pos[k][0], self.mlps_pos[k][1]
            if (self.fused and xyz.is_cuda and self.pool_method == 'max_pool' and feats_in.shape[1] <= 32 and pos_conv.bias is None
                    and not (torch.is_grad_enabled() and feats_in.dtype != torch.float32)):
                # device path: voxel query, then ONE kernel for group + position MLP + BatchNorm + ReLU + max
                # (csrc/voxel_roi_pool.hip); the CPU run below is the reference's op chain
                idx_raw = voxel_query_utils.voxel_query_raw(grouper.max_range, grouper.radius, grouper.nsample, xyz, new_xyz,
                                                            new_coords, voxel2point_indices)
                gamma = pos_bn.weight if pos_bn.affine else torch.ones(feats_in.shape[1], device=xyz.device)
                beta = pos_bn.bias if pos_bn.affine else torch.zeros(feats_in.shape[1], device=xyz.device)
                pooled, _ = _FusedVoxelRoIPool.apply(xyz, new_xyz, feats_in, idx_raw, pos_conv.weight.view(feats_in.shape[1], 3),
                                                     gamma, beta, pos_bn)                     # (C, M) channel-major
                per_scale.append(self.mlps_out[k](pooled.unsqueeze(0)).squeeze(0).permute(1, 0))
                continue
            grouped, grouped_xyz, empty = grouper(new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                                  feats_in, voxel2point_indices)
            keep = (~empty).view(-1, 1, 1).to(grouped.dtype)
            grouped = (grouped * keep).permute(1, 0, 2).unsqueeze(0)               # (1, C, M, nsample)
            rel_xyz = ((grouped_xyz - new_xyz.unsqueeze(-1)) * keep).permute(1, 0, 2).unsqueeze(0)
            x = self.relu(grouped + self.mlps_pos[k](rel_xyz))
            if self.pool_method == 'max_pool':
                x = x.max(dim=3).values                                            # (1, C, M)
            elif self.pool_method == 'avg_pool':
                x = x.mean(dim=3)
            else:
                raise NotImplementedError
            per_scale.append(self.mlps_out[k](x).squeeze(0).permute(1, 0))         # (M, C)
        return torch.cat(per_scale, dim=1)
