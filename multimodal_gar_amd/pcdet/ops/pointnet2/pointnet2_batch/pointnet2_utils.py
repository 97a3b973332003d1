"""Dense-batch PointNet++ primitives, (B, N, 3) coordinates / (B, C, N) features.

Mirror of the public surface of the reference's
pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py (same class / function names,
argument order, output shapes and dtypes, non-differentiable outputs), re-implemented on
top of the HIP kernels behind ``pointnet2_batch_cuda`` (libmgar_hip.so).

Ownership follows the reference: this layer allocates every output and scratch tensor
(zero-filled idx for ball_query :218, 1e10-filled ``temp`` for FPS :26, zero-filled grad
buffers :67,:146,:190); the native side only writes into them.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import pointnet2_batch_cuda as pointnet2


def _new(like: torch.Tensor, shape, dtype, fill=None) -> torch.Tensor:
    if fill is None:
        return torch.empty(shape, dtype=dtype, device=like.device)
    return torch.full(shape, fill, dtype=dtype, device=like.device)


PRUNED_FPS_MIN_POINTS = 8192   # below this the plain kernel is as fast once the Morton sort (~15 launches) is counted


class FarthestPointSampling(Function):
    """idx (B, npoint) int32 of an iterative farthest-point subset; first index is 0.
    Reference: pointnet2_utils.py:10-33 -> sampling_gpu.cu:101-216."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int) -> torch.Tensor:
        assert xyz.is_contiguous()
        batch, n_pts, _ = xyz.size()
        idx = _new(xyz, (batch, npoint), torch.int32)
        running_min = _new(xyz, (batch, n_pts), torch.float32, 1e10)
        if xyz.is_cuda and PRUNED_FPS_MIN_POINTS <= n_pts <= 16384 and hasattr(pointnet2, "farthest_point_sampling_pruned_wrapper"):
            pointnet2.farthest_point_sampling_pruned_wrapper(batch, n_pts, npoint, xyz, running_min, idx)
        elif xyz.is_cuda and 16384 < n_pts <= 65536 and hasattr(pointnet2, "farthest_point_sampling_buckets_wrapper"):
            pointnet2.farthest_point_sampling_buckets_wrapper(batch, n_pts, npoint, xyz, running_min, idx)
        else:
            pointnet2.farthest_point_sampling_wrapper(batch, n_pts, npoint, xyz, running_min, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_idx=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class GatherOperation(Function):
    """out[b, c, j] = features[b, c, idx[b, j]].  Reference: pointnet2_utils.py:39-73."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous()
        batch, npoint = idx.size()
        _, chans, n_pts = features.size()
        out = _new(features, (batch, chans, npoint), torch.float32)
        # a copy: a bf16 payload goes through fp32 and back bit-exactly (xyz, the usual operand, is fp32 anyway)
        pointnet2.gather_points_wrapper(batch, chans, n_pts, npoint, features.float(), idx, out)
        out = out.to(features.dtype)
        ctx.save_for_backward(idx)
        ctx.src_shape = (chans, n_pts)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        chans, n_pts = ctx.src_shape
        batch, npoint = idx.size()
        grad_features = _new(grad_out, (batch, chans, n_pts), torch.float32, 0.0)
        pointnet2.gather_points_grad_wrapper(batch, chans, n_pts, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """(dist, idx), both (B, n, 3): Euclidean distance to / index of the 3 nearest known
    points.  The kernel returns squared distances; the sqrt is taken here like the
    reference (pointnet2_utils.py:76-105)."""

    @staticmethod
    def forward(ctx, unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        assert unknown.is_contiguous() and known.is_contiguous()
        batch, n_unknown, _ = unknown.size()
        n_known = known.size(1)
        dist2 = _new(unknown, (batch, n_unknown, 3), torch.float32)
        idx = _new(unknown, (batch, n_unknown, 3), torch.int32)
        pointnet2.three_nn_wrapper(batch, n_unknown, n_known, unknown, known, dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist=None, grad_idx=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """out[b, c, i] = sum_k weight[b, i, k] * features[b, c, idx[b, i, k]].
    Reference: pointnet2_utils.py:108-153."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        batch, chans, n_known = features.size()
        n_unknown = idx.size(1)
        out = _new(features, (batch, chans, n_unknown), features.dtype)     # payload dtype in = payload dtype out
        pointnet2.three_interpolate_wrapper(batch, chans, n_known, n_unknown, features, idx, weight.float(), out)
        ctx.save_for_backward(idx, weight)
        ctx.n_known = n_known
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _three_interpolate_backward(grad_out, idx, weight, ctx.n_known), None, None


def _three_interpolate_backward(grad_out, idx, weight, n_known):
    """grad wrt the known features (B, C, m) from grad_out (B, C, n), which may be a CHANNEL SLICE of a wider tensor."""
    batch, chans, n_unknown = grad_out.size()
    grad_features = _new(grad_out, (batch, chans, n_known), torch.float32, 0.0)
    # the gradient of the decoder's torch.cat([interpolated, skip]) arrives as a CHANNEL SLICE of the wider tensor: the
    # device kernels read it in place (batch stride), where the reference copies it (up to 2 GB per launch at c3)
    bstride = None
    if grad_out.is_cuda and grad_out.dtype == torch.float32 and not grad_out.is_contiguous() and grad_out.stride(2) == 1 \
            and grad_out.stride(1) == n_unknown and grad_out.stride(0) >= chans * n_unknown \
            and (n_unknown % 4 != 0 or (grad_out.stride(0) % 4 == 0 and grad_out.data_ptr() % 16 == 0)) \
            and hasattr(pointnet2, "_sliced_ptr"):
        bstride = grad_out.stride(0)
    else:
        grad_out = grad_out.contiguous().float()
    m = n_known
    if grad_out.is_cuda and n_unknown <= 36864 and m <= 65535 and chans >= 16 and hasattr(pointnet2, "three_interpolate_grad_sorted_wrapper"):
        # inverted index: the 3n entries of every cloud sorted (stable) by known point, built once and
        # shared by all channels -- every known point is then summed by one owner, without atomics
        key = idx + (torch.arange(batch, device=idx.device, dtype=torch.int32) * m).view(-1, 1, 1) \
            if batch * m < 2 ** 31 else idx.long() + (torch.arange(batch, device=idx.device) * m).view(-1, 1, 1)
        order = torch.argsort(key.view(-1), stable=True)
        packed = (idx.view(-1)[order] << 16) | ((order // 3) % n_unknown).int()
        entries = torch.stack((packed, weight.reshape(-1)[order].view(torch.int32)), dim=1).contiguous()
        pointnet2.three_interpolate_grad_sorted_wrapper(batch, chans, n_unknown, m, grad_out, entries, grad_features, bstride)
    elif bstride is not None:
        pointnet2.three_interpolate_grad_wrapper(batch, chans, n_unknown, n_known, grad_out, idx, weight, grad_features, bstride)
    else:
        pointnet2.three_interpolate_grad_wrapper(batch, chans, n_unknown, n_known, grad_out, idx, weight, grad_features)
    return grad_features


class ThreeInterpolateConcat(Function):
    """torch.cat([three_interpolate(features, idx, weight), skip], dim=1) with the interpolation written straight into its
    channels of the result (csrc/interpolate.hip, mgar_three_interpolate_batch_into): the decoder step of the reference
    (PointnetFPModule.forward, pointnet2_batch/pointnet2_modules.py:139-148) without the pass that copies the interpolated
    half -- 2 GB of it at the finest level of config c3.  Device only."""

    @staticmethod
    def forward(ctx, features, idx, weight, skip):
        assert features.is_cuda and features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        batch, chans, n_known = features.size()
        n_unknown = idx.size(1)
        merged = _new(features, (batch, chans + skip.shape[1], n_unknown), features.dtype)
        pointnet2.three_interpolate_into_wrapper(batch, chans, n_known, n_unknown, features, idx, weight.float(), merged)
        merged[:, chans:].copy_(skip)
        ctx.save_for_backward(idx, weight)
        ctx.dims = (n_known, chans)
        return merged

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_merged):
        idx, weight = ctx.saved_tensors
        n_known, chans = ctx.dims
        grad_merged = grad_merged.contiguous()
        grad_features = _three_interpolate_backward(grad_merged[:, :chans], idx, weight, n_known) if ctx.needs_input_grad[0] else None
        return grad_features, None, None, grad_merged[:, chans:] if ctx.needs_input_grad[3] else None


three_interpolate = ThreeInterpolate.apply


class ThreeInterpolateAdd(Function):
    """base + three_interpolate(features, idx, weight), the interpolation added IN PLACE into ``base`` (a fresh tensor of the
    caller: the skip projection of PointnetFPModule's "project, then interpolate" path) -- csrc/interpolate.hip,
    mgar_three_interpolate_batch_add.  Device only."""

    @staticmethod
    def forward(ctx, features, idx, weight, base):
        assert features.is_cuda and features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous() and base.is_contiguous()
        batch, chans, n_known = features.size()
        n_unknown = idx.size(1)
        assert base.shape == (batch, chans, n_unknown) and base.dtype == features.dtype
        pointnet2.three_interpolate_add_wrapper(batch, chans, n_known, n_unknown, features, idx, weight.float(), base)
        ctx.mark_dirty(base)
        ctx.save_for_backward(idx, weight)
        ctx.n_known = n_known
        return base

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_features = _three_interpolate_backward(grad_out, idx, weight, ctx.n_known) if ctx.needs_input_grad[0] else None
        return grad_features, None, None, grad_out if ctx.needs_input_grad[3] else None


def three_interpolate_concat(features, idx, weight, skip):
    """cat([three_interpolate(features, idx, weight), skip], 1); on the device in one pass over the interpolated half."""
    if features.is_cuda and skip.dtype == features.dtype and hasattr(pointnet2, "three_interpolate_into_wrapper"):
        return ThreeInterpolateConcat.apply(features.contiguous(), idx, weight, skip)
    return torch.cat([three_interpolate(features, idx, weight), skip], dim=1)


class GroupingOperation(Function):
    """out[b, c, p, s] = features[b, c, idx[b, p, s]].  Reference: pointnet2_utils.py:156-197."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        assert features.is_contiguous() and idx.is_contiguous()
        batch, npoint, nsample = idx.size()
        _, chans, n_pts = features.size()
        out = _new(features, (batch, chans, npoint, nsample), torch.float32)
        pointnet2.group_points_wrapper(batch, chans, n_pts, npoint, nsample, features.float(), idx, out)   # a copy: exact for bf16
        out = out.to(features.dtype)
        ctx.save_for_backward(idx)
        ctx.n_pts = n_pts
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        batch, chans, npoint, nsample = grad_out.size()
        grad_features = _new(grad_out, (batch, chans, ctx.n_pts), torch.float32, 0.0)
        pointnet2.group_points_grad_wrapper(batch, chans, ctx.n_pts, npoint, nsample, grad_out.contiguous(), idx,
                                            grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """idx (B, npoint, nsample) int32: the first nsample point indices (ascending) closer
    than `radius` (strict) to each centre, padded with the first hit; all-zero row if the
    ball is empty.  Reference: pointnet2_utils.py:200-228 -> ball_query_gpu.cu:15-51."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        batch, n_pts, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = _new(xyz, (batch, npoint, nsample), torch.int32, 0)
        pointnet2.ball_query_wrapper(batch, n_pts, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_idx=None):
        return None, None, None, None


ball_query = BallQuery.apply


def ball_query_multi(radii, nsamples, xyz, new_xyz):
    """ball_query for several (radius, nsample) pairs over the same xyz / new_xyz in one scan of the cloud
    (csrc/ball_query.hip, ball_query_multi_kernel) -> list of idx tensors, each as BallQuery returns it."""
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    batch, n_pts, _ = xyz.size()
    npoint = new_xyz.size(1)
    out = [_new(xyz, (batch, npoint, ns), torch.int32, 0) for ns in nsamples]
    for g0 in range(0, len(out), 4):
        pointnet2.ball_query_multi_wrapper(batch, n_pts, npoint, list(radii[g0:g0 + 4]), list(nsamples[g0:g0 + 4]), new_xyz, xyz,
                                           out[g0:g0 + 4])
    return out


class _FusedQueryGroup(Function):
    """Relative xyz (3 rows) + grouped features (C rows) written once by one kernel
    (csrc/query_group.hip) instead of transpose / group / subtract / group / cat."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx):
        batch, n_pts, _ = xyz.size()
        npoint, nsample = idx.size(1), idx.size(2)
        chans = 0 if features is None else features.size(1)
        out = _new(xyz, (batch, 3 + chans, npoint, nsample), torch.float32 if features is None else features.dtype)
        pointnet2.query_group_wrapper(batch, chans, n_pts, npoint, nsample, xyz, new_xyz,
                                      None if features is None else features.contiguous(), idx, out)
        ctx.save_for_backward(idx)
        ctx.dims = (chans, n_pts)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        chans, n_pts = ctx.dims
        if chans == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None
        batch, _, npoint, nsample = grad_out.size()
        grad_features = _new(grad_out, (batch, chans, n_pts), torch.float32, 0.0)
        pointnet2.query_group_grad_wrapper(batch, chans, n_pts, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return None, None, grad_features, None


class _FusedQueryGroupProj(Function):
    """y = gather(zf) + wx . rel_xyz  (csrc/query_group.hip, "project, then group"): the output of
    a first shared-MLP layer whose feature half (zf = W_f features) was applied before grouping."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, zf, wx, idx, rows_bwd=False):
        """rows_bwd: the consumer hands the gradient back as rows (nn_utils.forward_maxpool(rowmajor_input_grad=True)); the
        backward then takes the atomic-free owner-computes kernels of the stacked layout (a dense batch IS a stacked batch with
        equal counts), which also form d wx from the coordinates -- no relative coordinates are stored for it."""
        batch, n_pts, _ = xyz.size()
        npoint, nsample = idx.size(1), idx.size(2)
        chans = zf.size(1)
        zf, wx = zf.contiguous(), wx.contiguous().float()
        rows = bool(rows_bwd) and xyz.is_cuda and chans <= 64 and zf.dtype == torch.float32 and n_pts <= 262144 \
            and batch * npoint * nsample < 2 ** 31
        # rel (relative coordinates) is only read by the backward (d wx): a forward-only call does not write it
        rel = _new(xyz, (batch, 3, npoint, nsample), zf.dtype) if any(ctx.needs_input_grad) and not rows else None
        y = _new(xyz, (batch, chans, npoint, nsample), zf.dtype)
        pointnet2.query_group_proj_wrapper(batch, chans, n_pts, npoint, nsample, xyz, new_xyz, zf, wx, idx, rel, y)
        if rows:
            ctx.save_for_backward(idx, xyz, new_xyz)
        else:
            ctx.save_for_backward(idx, rel)
        ctx.rows = rows
        ctx.dims = (chans, n_pts)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        chans, n_pts = ctx.dims
        batch, _, npoint, nsample = grad_y.size()
        if ctx.rows:
            # round 3: the LDS-atomic row accumulation of csrc/query_group.hip (qg_batch_bwd_lds_kernel) ran at the pace of the
            # LDS float-atomic unit (4.3 ms per c3 step); the stacked layout's inverted index + owner-computes rows kernel has no
            # atomics, a fixed summation order, and reads the gradient as rows -- which is how the MLP's first BatchNorm
            # backward hands it over (bn_ops._bwd_rowmajor): no transposing copy
            from ..pointnet2_stack import pointnet2_stack_cuda as S
            idx, xyz, new_xyz = ctx.saved_tensors
            cols = batch * npoint * nsample
            gy_t = grad_y.permute(0, 2, 3, 1)                                   # (B, M, ns, C): contiguous iff the memory is rows
            gy_t = (gy_t if gy_t.is_contiguous() else gy_t.contiguous()).view(cols, chans)
            grad_rows = torch.zeros((batch * n_pts, chans), dtype=torch.float32, device=grad_y.device)
            qcnt = torch.full((batch,), npoint, dtype=torch.int32, device=grad_y.device)
            pcnt = torch.full((batch,), n_pts, dtype=torch.int32, device=grad_y.device)
            grad_wx = S.query_group_proj_grad_rows_wrapper(batch, batch * npoint, chans, nsample, gy_t, idx.view(-1, nsample), qcnt, pcnt,
                                                           grad_rows, xyz=xyz.view(-1, 3), new_xyz=new_xyz.view(-1, 3))
            return None, None, grad_rows.view(batch, n_pts, chans).transpose(1, 2), grad_wx, None, None
        idx, rel = ctx.saved_tensors
        grad_y = grad_y.contiguous()
        grad_zf = _new(grad_y, (batch, chans, n_pts), torch.float32, 0.0)
        pointnet2.query_group_proj_grad_wrapper(batch, chans, n_pts, npoint, nsample, grad_y, idx, grad_zf)
        from .....nn_utils import pointwise_dw
        grad_wx = pointwise_dw(rel.flatten(2), grad_y.flatten(2))                              # (C, 3)
        return None, None, grad_zf, grad_wx, None, None


class QueryAndGroup(nn.Module):
    """ball_query -> group xyz (made relative to the centre) -> group features -> concat.
    Output (B, 3 + C, npoint, nsample).  Reference: pointnet2_utils.py:231-264."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None, idx=None):
        """``idx``: this ball query's result computed ahead by the caller (the scales of an MSG module share one scan)."""
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        if not self.use_xyz:
            return grouping_operation(features, idx)
        return _FusedQueryGroup.apply(xyz, new_xyz, features, idx)

    def forward_projected(self, xyz, new_xyz, features, weight, idx=None, rows_bwd=False):
        """First shared-MLP layer folded into the grouping: returns  W [rel_xyz ; grouped features]
        (B, C_out, npoint, nsample) for a bias-free point-wise conv weight (C_out, 3 + C), without
        ever building the (3 + C)-channel grouped tensor."""
        assert self.use_xyz and features is not None
        w = weight.view(weight.shape[0], -1)
        zf = torch.bmm(w[:, 3:].unsqueeze(0).expand(features.shape[0], -1, -1), features)     # (B, C_out, N)
        if idx is None:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        return _FusedQueryGroupProj.apply(xyz, new_xyz, zf, w[:, :3], idx, rows_bwd)


class GroupAll(nn.Module):
    """One group holding every point: (B, 3 + C, 1, N).  Reference: pointnet2_utils.py:267-290."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: torch.Tensor, new_xyz: torch.Tensor, features: Optional[torch.Tensor] = None):
        all_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return all_xyz
        all_feats = features.unsqueeze(2)
        return torch.cat([all_xyz, all_feats], dim=1) if self.use_xyz else all_feats
