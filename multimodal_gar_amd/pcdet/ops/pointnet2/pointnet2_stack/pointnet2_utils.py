"""Stacked-batch PointNet++ primitives: (N1+N2+..., 3|C) rows plus per-sample counts.

Mirror of the public surface of the reference's
pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:1-303 (ball_query, grouping_operation,
QueryAndGroup, farthest_point_sample, stack_farthest_point_sample, three_nn,
three_interpolate), on top of the HIP kernels behind ``pointnet2_stack_cuda``.
The vector-pool family (:306-457) is out of scope (PV-RCNN++ only).
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import pointnet2_stack_cuda as pointnet2


def _empty(like, shape, dtype):
    return torch.empty(shape, dtype=dtype, device=like.device)


class BallQuery(Function):
    """Returns (idx (M, nsample) int32 LOCAL indices, empty_ball_mask (M) bool).
    The kernel marks an empty ball with idx[row, 0] == -1; such rows are reported in the
    mask and reset to 0 (reference pointnet2_utils.py:33-37)."""

    @staticmethod
    def forward(ctx, radius: float, nsample: int, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor,
                new_xyz: torch.Tensor, new_xyz_batch_cnt):
        assert new_xyz.is_contiguous() and new_xyz_batch_cnt.is_contiguous()
        assert xyz.is_contiguous() and xyz_batch_cnt.is_contiguous()
        n_samples = xyz_batch_cnt.shape[0]
        n_query = new_xyz.shape[0]
        idx = torch.zeros((n_query, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(n_samples, n_query, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz,
                                     xyz_batch_cnt, idx)
        empty_ball_mask = idx[:, 0] == -1
        idx[empty_ball_mask] = 0
        ctx.mark_non_differentiable(idx, empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    """out[m, c, s] = features[start(sample of m) + idx[m, s], c]  -> (M, C, nsample).
    Reference: pointnet2_utils.py:52-106."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, features_batch_cnt: torch.Tensor,
                idx: torch.Tensor, idx_batch_cnt: torch.Tensor):
        assert features.is_contiguous() and features_batch_cnt.is_contiguous()
        assert idx.is_contiguous() and idx_batch_cnt.is_contiguous()
        assert features.shape[0] == features_batch_cnt.sum(), \
            'features: %s, features_batch_cnt: %s' % (str(features.shape), str(features_batch_cnt))
        assert idx.shape[0] == idx_batch_cnt.sum(), \
            'idx: %s, idx_batch_cnt: %s' % (str(idx.shape), str(idx_batch_cnt))
        n_query, nsample = idx.size()
        n_rows, chans = features.size()
        n_samples = idx_batch_cnt.shape[0]
        out = _empty(features, (n_query, chans, nsample), torch.float32)
        # the un-fused grouping copies elements: a bf16 payload goes through fp32 and back, bit-exactly (the fused
        # query-and-group kernels, which carry the traffic, read and write bf16 directly)
        pointnet2.group_points_wrapper(n_samples, n_query, chans, nsample, features.float(), features_batch_cnt, idx,
                                       idx_batch_cnt, out)
        out = out.to(features.dtype)
        ctx.save_for_backward(idx, features_batch_cnt, idx_batch_cnt)
        ctx.dims = (n_samples, n_rows)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: torch.Tensor):
        idx, features_batch_cnt, idx_batch_cnt = ctx.saved_tensors
        n_samples, n_rows = ctx.dims
        n_query, chans, nsample = grad_out.size()
        grad_features = torch.zeros((n_rows, chans), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(n_samples, n_query, chans, n_rows, nsample, grad_out.contiguous(), idx,
                                            idx_batch_cnt, features_batch_cnt, grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class _FusedQueryGroup(Function):
    """ball query + relative xyz + grouped features, zeroed for empty balls, written once as a
    CHANNEL-MAJOR (3 + C, M * nsample) tensor (csrc/query_group.hip).  Returns (grouped, idx_raw)
    where idx_raw[row, 0] == -1 marks an empty ball."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features):
        n_samples = xyz_batch_cnt.shape[0]
        n_query = new_xyz.shape[0]
        idx = torch.zeros((n_query, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(n_samples, n_query, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        chans = 0 if features is None else features.shape[1]
        out = _empty(xyz, (3 + chans, n_query * nsample), torch.float32 if features is None else features.dtype)
        pointnet2.query_group_wrapper(n_samples, n_query, chans, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                      None if features is None else features.contiguous(), idx, out)
        ctx.save_for_backward(idx, xyz_batch_cnt, new_xyz_batch_cnt)
        ctx.dims = (n_samples, n_query, chans, nsample, 0 if features is None else features.shape[0])
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, grad_idx=None):
        idx, xyz_batch_cnt, new_xyz_batch_cnt = ctx.saved_tensors
        n_samples, n_query, chans, nsample, n_rows = ctx.dims
        if chans == 0 or not ctx.needs_input_grad[6]:
            return (None,) * 7
        grad_features = torch.zeros((n_rows, chans), dtype=torch.float32, device=grad_out.device)
        pointnet2.query_group_grad_wrapper(n_samples, n_query, chans, nsample, grad_out.contiguous(), idx,
                                           new_xyz_batch_cnt, xyz_batch_cnt, grad_features)
        return None, None, None, None, None, None, grad_features


class _FusedQueryGroupProj(Function):
    """ball query + y = gather(zf) + wx . rel_xyz, channel-major (C_out, M * nsample), zero columns for
    empty balls ("project, then group", csrc/query_group.hip)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, zf, wx):
        n_samples = xyz_batch_cnt.shape[0]
        n_query = new_xyz.shape[0]
        idx = torch.zeros((n_query, nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(n_samples, n_query, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        zf, wx = zf.contiguous(), wx.contiguous().float()
        chans = zf.shape[1]
        # rel (the relative coordinates) is only needed by the backward (d wx): a forward-only call does not write it
        rel = _empty(xyz, (3, n_query * nsample), zf.dtype) if any(ctx.needs_input_grad) else None
        y = _empty(xyz, (chans, n_query * nsample), zf.dtype)
        pointnet2.query_group_proj_wrapper(n_samples, n_query, chans, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                           zf, wx, idx, rel, y)
        ctx.save_for_backward(idx, xyz_batch_cnt, new_xyz_batch_cnt, rel)
        ctx.dims = (n_samples, n_query, chans, nsample, zf.shape[0])
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y, grad_idx=None):
        idx, xyz_batch_cnt, new_xyz_batch_cnt, rel = ctx.saved_tensors
        n_samples, n_query, chans, nsample, n_rows = ctx.dims
        grad_y = grad_y.contiguous()
        grad_zf = torch.zeros((n_rows, chans), dtype=torch.float32, device=grad_y.device)
        pointnet2.query_group_proj_grad_wrapper(n_samples, n_query, chans, nsample, grad_y, idx, new_xyz_batch_cnt,
                                                xyz_batch_cnt, grad_zf)
        from .....nn_utils import pointwise_dw
        grad_wx = pointwise_dw(rel.unsqueeze(0), grad_y.unsqueeze(0))                          # (C, 3)
        return None, None, None, None, None, None, grad_zf, grad_wx


class _FusedQueryGroupProjMSG(Function):
    """Multi-scale "project, then group": the scales of one StackSAModuleMSG share the source
    features, so their first-layer projections are ONE GEMM zf = features @ [W_f,1; W_f,2; ...]^T
    (N, sum C_k) -- one pass over the features instead of one per scale, and one pass for each of the
    two gradients.  Per scale k: ball query + y_k = gather(zf[:, cols_k]) + wx_k . rel_xyz
    (csrc/query_group.hip, zf read with leading dimension sum C_k).  The weight gradient
    d W_f = grad_zf^T features runs on csrc/rowmajor_dw.hip.

    apply(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, radii, nsamples, rows_bwd, *weights)
    with weights[k] (C_k, 3 + C) -> (y_1, ..., y_K, s_1, ..., s_K), y_k (C_k, M * nsample_k) channel-major, s_k the BatchNorm
    statistics partials of y_k left by the grouping kernel (bn_ops.stats_partial_buffer; an empty tensor where that does not
    apply), so that the BatchNorm after the folded layer needs no pass over y_k.
    rows_bwd[k]: the consumer of y_k hands its gradient back as rows (M * nsample_k, C_k) (nn_utils.forward_maxpool(
    rowmajor_input_grad=True)); scale k then takes the atomic-free, bit-reproducible backward (qg_stack_bwd_rows_kernel),
    which also forms d wx_k from the coordinates: the forward stores no relative coordinates for it."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, radii, nsamples, rows_bwd, *weights):
        """features: (N, C) stacked rows, or (B, C, n) CHANNEL-MAJOR with n points in every sample (the layout the
        PointNet++ trunk produces: no transposed copy of the feature matrix is needed for the projection GEMM)."""
        n_samples, n_query = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        features = features.contiguous()
        channel_major = features.dim() == 3
        ws = [w.reshape(w.shape[0], -1) for w in weights]
        chans = [w.shape[0] for w in ws]
        ld = sum(chans)
        w_f = torch.cat([w[:, 3:] for w in ws], 0).contiguous()                                # (ld, C)
        if channel_major:
            b, c, n = features.shape
            assert b == n_samples and b * n == xyz.shape[0]
            zf = torch.bmm(features.transpose(1, 2), w_f.t().unsqueeze(0).expand(b, c, ld)).view(b * n, ld)
        else:
            zf = features @ w_f.t()                                                            # (N, ld)
        outs, saved = [], []
        col = 0
        need_bwd = any(ctx.needs_input_grad)
        zf = zf.contiguous()
        # One scan per radius here.  (The multi-radius kernel, csrc/ball_query.hip, wins where the rows fill up and
        # the scans stop early -- the trunk's FPS centres: 2.56 vs 3.38 ms; the RoI grid points rarely fill their
        # smallest ball, every scan runs to the end of the cloud and the heavier per-pair path loses: 4.16 vs 3.51 ms.)
        idxs = [torch.zeros((n_query, ns), dtype=torch.int32, device=xyz.device) for ns in nsamples]
        # round 3: large clouds are binned into a cell grid ONCE and every radius queries it (csrc/ball_query_grid.hip):
        # ~130 candidates per query instead of the whole cloud
        from ..... import point_grid as G
        grid = None
        if xyz.is_cuda and n_query > 0 and G.wanted(xyz.shape[0] // max(n_samples, 1), nsamples) and hasattr(pointnet2, "ball_query_grid_wrapper"):
            grid = G.PointGrid(xyz, G.cell_for(radii), xyz_batch_cnt.int())
        for radius, nsample, idx in zip(radii, nsamples, idxs):
            if grid is not None:
                pointnet2.ball_query_grid_wrapper(n_samples, n_query, radius, nsample, new_xyz, new_xyz_batch_cnt, grid, idx)
            else:
                pointnet2.ball_query_wrapper(n_samples, n_query, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        rows_bwd = tuple(bool(r) and c <= 64 and zf.dtype == torch.float32 for r, c in zip(rows_bwd, chans))
        from .....bn_ops import stats_partial_buffer
        stats = []
        for radius, nsample, w, c, idx, rows in zip(radii, nsamples, ws, chans, idxs, rows_bwd):
            wx = w[:, :3].contiguous().float()
            rel = _empty(xyz, (3, n_query * nsample), zf.dtype) if need_bwd and not rows else None
            y = _empty(xyz, (c, n_query * nsample), zf.dtype)
            st = stats_partial_buffer(y, c, n_query * nsample)        # None off the device / for bf16 payloads / odd sizes
            pointnet2.query_group_proj_wrapper(n_samples, n_query, c, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                               zf, wx, idx, rel, y, zf_ld=ld, zf_col=col, **({"out_stats": st} if st is not None else {}))
            stats.append(st if st is not None else y.new_empty((0,)))
            outs.append(y)
            saved += [idx, rel if rel is not None else idx]
            col += c
        ctx.save_for_backward(xyz_batch_cnt, new_xyz_batch_cnt, features, w_f, xyz, new_xyz, *saved)
        ctx.meta = (n_samples, n_query, tuple(chans), tuple(nsamples), tuple(w.shape for w in weights), rows_bwd)
        ctx.mark_non_differentiable(*stats)
        return tuple(outs) + tuple(stats)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        from .....nn_utils import pointwise_dw
        grad_ys = grads[:len(grads) // 2]
        xyz_batch_cnt, new_xyz_batch_cnt, features, w_f, xyz, new_xyz = ctx.saved_tensors[:6]
        saved = ctx.saved_tensors[6:]
        n_samples, n_query, chans, nsamples, w_shapes, rows_bwd = ctx.meta
        ld = sum(chans)
        channel_major = features.dim() == 3
        n_rows = features.shape[0] * features.shape[2] if channel_major else features.shape[0]
        n_feat = features.shape[1]
        grad_zf = torch.zeros((n_rows, ld), dtype=torch.float32, device=features.device)
        grad_wx = []
        col = 0
        for k, (c, nsample) in enumerate(zip(chans, nsamples)):
            idx, rel = saved[2 * k], saved[2 * k + 1]
            gy = grad_ys[k]
            if rows_bwd[k]:
                # rows (M * nsample, C_k): as handed back by the MLP's first BatchNorm backward, or (any other producer) one
                # transposing copy; owner-computes gather over the inverted index, no atomics, bit-reproducible
                gy_t = gy.t() if gy.stride() == (1, c) else gy.t().contiguous()
                grad_wx.append(pointnet2.query_group_proj_grad_rows_wrapper(
                    n_samples, n_query, c, nsample, gy_t, idx, new_xyz_batch_cnt, xyz_batch_cnt, grad_zf, zf_ld=ld, zf_col=col,
                    xyz=xyz, new_xyz=new_xyz))                                                 # (C_k, 3)
                col += c
                continue
            gy = gy.contiguous()
            pointnet2.query_group_proj_grad_wrapper(n_samples, n_query, c, nsample, gy, idx, new_xyz_batch_cnt, xyz_batch_cnt,
                                                    grad_zf, zf_ld=ld, zf_col=col)
            grad_wx.append(pointwise_dw(rel.unsqueeze(0), gy.unsqueeze(0)))                    # (C_k, 3)
            col += c
        if channel_major:
            b, c, n = features.shape
            g3 = grad_zf.view(b, n, ld)
            grad_features = torch.bmm(w_f.t().unsqueeze(0).expand(b, c, ld), g3.transpose(1, 2)) if ctx.needs_input_grad[4] else None
            grad_wf = torch.bmm(g3.transpose(1, 2), features.transpose(1, 2)).sum(0)           # (ld, C)
        else:
            grad_features = grad_zf @ w_f if ctx.needs_input_grad[4] else None
            if ld <= 96 and n_feat <= 128:
                grad_wf = pointnet2.rowmajor_dw(grad_zf, features)                             # (ld, C)
            else:
                grad_wf = grad_zf.t() @ features
        grad_ws, col = [], 0
        for c, gwx, shape in zip(chans, grad_wx, w_shapes):
            grad_ws.append(torch.cat([gwx, grad_wf[col:col + c]], 1).view(shape))
            col += c
        return (None, None, None, None, grad_features, None, None, None, *grad_ws)


class QueryAndGroup(nn.Module):
    """Returns (new_features (M, 3 + C, nsample), idx).  Relative xyz and features of empty
    balls are zeroed.  Reference: pointnet2_utils.py:112-159."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: torch.Tensor, xyz_batch_cnt: torch.Tensor,
                new_xyz: torch.Tensor, new_xyz_batch_cnt: torch.Tensor,
                features: torch.Tensor = None):
        assert xyz.shape[0] == xyz_batch_cnt.sum(), \
            'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_xyz.shape[0] == new_xyz_batch_cnt.sum(), \
            'new_xyz: %s, new_xyz_batch_cnt: %s' % (str(new_xyz.shape), str(new_xyz_batch_cnt))
        grouped, idx_raw = self.forward_channel_major(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)
        # reference layout (M, C', nsample) as a view of the channel-major tensor; cleaned indices
        new_features = grouped.view(grouped.shape[0], new_xyz.shape[0], self.nsample).permute(1, 0, 2)
        idx = idx_raw.masked_fill((idx_raw[:, :1] == -1), 0)
        return new_features, idx

    def forward_channel_major(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        """-> (grouped (3 + C | C, M * nsample) channel-major, idx_raw (M, nsample) with the kernel's
        -1 marker in column 0 of empty balls).  This is the layout the shared MLP consumes."""
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        grouped, idx_raw = _FusedQueryGroup.apply(self.radius, self.nsample, xyz, xyz_batch_cnt.int(), new_xyz,
                                                  new_xyz_batch_cnt.int(), features)
        return (grouped if self.use_xyz else grouped[3:]), idx_raw

    def forward_projected(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, weight):
        """First shared-MLP layer folded into the grouping: W [rel_xyz ; grouped features] as a
        channel-major (C_out, M * nsample) tensor for a bias-free point-wise conv weight
        (C_out, 3 + C); the (3 + C)-channel grouped tensor is never built."""
        assert self.use_xyz and features is not None
        w = weight.view(weight.shape[0], -1)
        zf = features @ w[:, 3:].t()                                                          # (N, C_out)
        return _FusedQueryGroupProj.apply(self.radius, self.nsample, xyz, xyz_batch_cnt.int(), new_xyz,
                                          new_xyz_batch_cnt.int(), zf, w[:, :3])


class FarthestPointSampling(Function):
    """Dense-batch FPS re-exported by the stack package (reference pointnet2_utils.py:162-188)."""

    @staticmethod
    def forward(ctx, xyz: torch.Tensor, npoint: int):
        assert xyz.is_contiguous()
        batch, n_pts, _ = xyz.size()
        idx = _empty(xyz, (batch, npoint), torch.int32)
        running_min = torch.full((batch, n_pts), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(batch, n_pts, npoint, xyz, running_min, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class StackFarthestPointSampling(Function):
    """FPS per stacked segment; returns GLOBAL row ids, concatenated (sum of npoint).
    Reference: pointnet2_utils.py:191-225 -> sampling_gpu.cu:188-348."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, npoint):
        assert xyz.is_contiguous() and xyz.shape[1] == 3
        n_samples = len(xyz_batch_cnt)
        if not isinstance(npoint, torch.Tensor):
            if not isinstance(npoint, list):
                npoint = [npoint] * n_samples
            npoint = torch.tensor(npoint, device=xyz.device).int()
        npoint = npoint.to(device=xyz.device, dtype=torch.int32).contiguous()
        running_min = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
        idx = _empty(xyz, (int(npoint.sum().item()),), torch.int32)
        pointnet2.stack_farthest_point_sampling_wrapper(xyz, running_min, xyz_batch_cnt.int().contiguous(), idx,
                                                        npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


stack_farthest_point_sample = StackFarthestPointSampling.apply


class ThreeNN(Function):
    """(dist (N, 3), idx (N, 3) GLOBAL rows of `known`).  Reference: pointnet2_utils.py:228-258."""

    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        assert unknown.dim() == 2 and unknown.shape[1] == 3
        assert known.dim() == 2 and known.shape[1] == 3
        assert len(unknown_batch_cnt) == len(known_batch_cnt)
        dist2 = unknown.new_zeros(unknown.shape)
        idx = torch.zeros(unknown.shape, dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(unknown.contiguous(), unknown_batch_cnt.contiguous(), known.contiguous(),
                                   known_batch_cnt.contiguous(), dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """out[i, c] = sum_k weight[i, k] * features[idx[i, k], c]  -> (N, C).
    Reference: pointnet2_utils.py:264-300."""

    @staticmethod
    def forward(ctx, features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor):
        assert idx.shape[0] == weight.shape[0] and idx.shape[1] == weight.shape[1] == 3
        idx, weight = idx.contiguous(), weight.contiguous()
        ctx.save_for_backward(idx, weight)
        ctx.n_known = features.shape[0]
        out = features.new_zeros((idx.shape[0], features.shape[1]))
        pointnet2.three_interpolate_wrapper(features.contiguous(), idx, weight.float(), out)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out: torch.Tensor):
        idx, weight = ctx.saved_tensors
        grad_features = grad_out.new_zeros((ctx.n_known, grad_out.shape[1]))
        pointnet2.three_interpolate_grad_wrapper(grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply
