"""CPU execution backend built on the oracle -- for bench.py's ``cpu_baseline`` leg and for
tests ONLY (see oracle/mgar_oracle.c header).

``use_cpu_oracle()`` is a context manager that temporarily replaces the functions of the two
native shim modules (pointnet2_batch_cuda / pointnet2_stack_cuda) and the three fused-op entry
points with CPU implementations: the C oracle for the pointnet2 ops, plain fp32 torch for
RoIAlign / DAFM attention / GATv2.  Inside the context the SAME Python model code runs on CPU
tensors, which is how "the reference CPU path"  is timed.  The product never
enters this context; outside it every op refuses CPU tensors.
"""
import contextlib
import ctypes
import math

import torch

from . import oracle as O


def _p(t):
    assert t.device.type == "cpu" and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def _cf(x):
    return ctypes.c_float(float(x))


class _Batch:
    @staticmethod
    def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
        O.lib().orc_ball_query_batch(b, n, m, _cf(radius), nsample, _p(new_xyz), _p(xyz), _p(idx)); return 1

    @staticmethod
    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        O.lib().orc_group_points_batch(b, c, n, npoints, nsample, _p(points), _p(idx), _p(out)); return 1

    @staticmethod
    def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
        O.lib().orc_group_points_grad_batch(b, c, n, npoints, nsample, _p(grad_out), _p(idx), _p(grad_points)); return 1

    @staticmethod
    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        O.lib().orc_gather_points(b, c, n, npoints, _p(points), _p(idx), _p(out)); return 1

    @staticmethod
    def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
        O.lib().orc_gather_points_grad(b, c, n, npoints, _p(grad_out), _p(idx), _p(grad_points)); return 1

    @staticmethod
    def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
        O.lib().orc_fps_batch(b, n, m, _p(points), _p(temp), _p(idx)); return 1

    @staticmethod
    def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
        O.lib().orc_three_nn_batch(b, n, m, _p(unknown), _p(known), _p(dist2), _p(idx)); return 1

    @staticmethod
    def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
        O.lib().orc_three_interpolate_batch(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out)); return 1

    @staticmethod
    def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
        O.lib().orc_three_interpolate_grad_batch(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(grad_points)); return 1


def _batch_query_group(b, c, n, npoints, nsample, xyz, new_xyz, features, idx, out):
    xyz_t = xyz.transpose(1, 2).contiguous()
    g = torch.empty((b, 3, npoints, nsample)); _Batch.group_points_wrapper(b, 3, n, npoints, nsample, xyz_t, idx, g)
    out[:, :3] = g - new_xyz.transpose(1, 2).unsqueeze(-1)
    if c:
        gf = torch.empty((b, c, npoints, nsample)); _Batch.group_points_wrapper(b, c, n, npoints, nsample, features, idx, gf)
        out[:, 3:] = gf
    return 1


def _batch_query_group_grad(b, c, n, npoints, nsample, grad_out, idx, grad_features):
    _Batch.group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out[:, 3:].contiguous(), idx, grad_features)
    return 1


_Batch.query_group_wrapper = staticmethod(_batch_query_group)
_Batch.query_group_grad_wrapper = staticmethod(_batch_query_group_grad)


class _Stack:
    @staticmethod
    def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
        O.lib().orc_ball_query_stack(B, M, _cf(radius), nsample, _p(new_xyz), _p(new_xyz_batch_cnt), _p(xyz),
                                     _p(xyz_batch_cnt), _p(idx)); return 1

    @staticmethod
    def voxel_query_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                            point_indices, idx):
        O.lib().orc_voxel_query(M, R1, R2, R3, nsample, _cf(radius), z_range, y_range, x_range, _p(new_xyz), _p(xyz),
                                _p(new_coords), _p(point_indices), _p(idx)); return 1

    farthest_point_sampling_wrapper = _Batch.farthest_point_sampling_wrapper

    @staticmethod
    def stack_farthest_point_sampling_wrapper(points, temp, xyz_batch_cnt, idx, num_sampled_points):
        O.lib().orc_fps_stack(xyz_batch_cnt.shape[0], _p(points), _p(temp), _p(xyz_batch_cnt), _p(idx),
                              _p(num_sampled_points)); return 1

    @staticmethod
    def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
        O.lib().orc_group_points_stack(B, M, C, nsample, _p(features), _p(features_batch_cnt), _p(idx),
                                       _p(idx_batch_cnt), _p(out)); return 1

    @staticmethod
    def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
        O.lib().orc_group_points_grad_stack(B, M, C, N, nsample, _p(grad_out), _p(idx), _p(idx_batch_cnt),
                                            _p(features_batch_cnt), _p(grad_features)); return 1

    @staticmethod
    def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
        O.lib().orc_three_nn_stack(unknown_batch_cnt.shape[0], unknown.shape[0], _p(unknown), _p(unknown_batch_cnt),
                                   _p(known), _p(known_batch_cnt), _p(dist2), _p(idx))

    @staticmethod
    def three_interpolate_wrapper(features, idx, weight, out):
        O.lib().orc_three_interpolate_stack(idx.shape[0], features.shape[1], _p(features), _p(idx), _p(weight), _p(out))

    @staticmethod
    def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
        O.lib().orc_three_interpolate_grad_stack(idx.shape[0], grad_out.shape[1], _p(grad_out), _p(idx), _p(weight),
                                                 _p(grad_features))


def _stack_query_group(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx_raw, out):
    empty = idx_raw[:, 0] == -1
    idx = idx_raw.masked_fill(empty[:, None], 0).contiguous()
    keep = (~empty).view(-1, 1, 1).float()
    g = torch.empty((M, 3, nsample)); _Stack.group_points_wrapper(B, M, 3, nsample, xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt, g)
    out[:3] = ((g - new_xyz.unsqueeze(-1)) * keep).permute(1, 0, 2).reshape(3, -1)
    if C:
        gf = torch.empty((M, C, nsample))
        _Stack.group_points_wrapper(B, M, C, nsample, features, xyz_batch_cnt, idx, new_xyz_batch_cnt, gf)
        out[3:] = (gf * keep).permute(1, 0, 2).reshape(C, -1)
    return 1


def _stack_query_group_grad(B, M, C, nsample, grad_out, idx_raw, new_xyz_batch_cnt, xyz_batch_cnt, grad_features):
    empty = idx_raw[:, 0] == -1
    idx = idx_raw.masked_fill(empty[:, None], 0).contiguous()
    g = grad_out[3:].view(C, M, nsample).permute(1, 0, 2) * (~empty).view(-1, 1, 1).float()
    _Stack.group_points_grad_wrapper(B, M, C, grad_features.shape[0], nsample, g.contiguous(), idx, new_xyz_batch_cnt,
                                     xyz_batch_cnt, grad_features)
    return 1


_Stack.query_group_wrapper = staticmethod(_stack_query_group)
_Stack.query_group_grad_wrapper = staticmethod(_stack_query_group_grad)


# ------------------------- fused ops: plain fp32 torch on CPU (autograd by torch) -------------------------
def roi_align_cpu(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    from multimodal_gar_amd.vision_ops import convert_boxes_to_roi_format
    rois = boxes if torch.is_tensor(boxes) else convert_boxes_to_roi_format(boxes)
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    if not input.requires_grad:
        n, c, h, w = input.shape
        out = torch.empty((rois.shape[0], c, ph, pw), dtype=torch.float32)
        inp = input.contiguous().float(); r = rois.contiguous().float()
        O.lib().orc_roi_align_fwd(_p(inp), n, c, h, w, _p(r), r.shape[0], ph, pw, _cf(spatial_scale), int(sampling_ratio),
                                  int(bool(aligned)), _p(out))
        return out
    raise NotImplementedError("CPU baseline only needs RoIAlign on the frozen I3D features")


def dafm_attention_cpu(q, k, v, de_flat, scene_off, de_off, sigma, scale):
    so, do = scene_off.tolist(), de_off.tolist()
    outs, atts = [], []
    for s in range(len(so) - 1):
        r0, r1 = so[s], so[s + 1]
        n = r1 - r0
        de = de_flat[do[s]:do[s] + n * n].view(n, n)
        e = torch.softmax(-(de / sigma), dim=1)
        att = torch.softmax((q[r0:r1] @ k[r0:r1].T) * e * scale, dim=1)
        outs.append(att @ v[r0:r1]); atts.append(att.reshape(-1))
    return torch.cat(outs), torch.cat(atts)


class _GatAggregateCpu:
    @staticmethod
    def apply(xl, xr, att, rowptr, col, edge_scale, heads, slope, by_source=None):
        n = xl.shape[0]
        c = xl.shape[1] // heads
        xl3, xr3 = xl.view(n, heads, c), xr.view(n, heads, c)
        deg = (rowptr[1:] - rowptr[:-1]).long()
        dst = torch.repeat_interleave(torch.arange(n), deg)
        src = col.long()
        z = torch.nn.functional.leaky_relu(xl3[src] + xr3[dst], slope)
        e = (z * att.view(1, heads, c)).sum(-1)                                # (E, H)
        emax = torch.full((n, heads), -math.inf).scatter_reduce(0, dst[:, None].expand(-1, heads), e, "amax")
        p = torch.exp(e - emax[dst])
        den = torch.zeros((n, heads)).index_add_(0, dst, p)
        alpha = p / den[dst]
        am = alpha if edge_scale is None else alpha * edge_scale
        out = torch.zeros((n, heads, c)).index_add_(0, dst, am[:, :, None] * xl3[src])
        return out.view(n, heads * c), alpha


def roipoint_pool3d_forward_cpu(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    O.lib().orc_roipoint_pool3d(xyz.shape[0], xyz.shape[1], boxes3d.shape[1], pts_feature.shape[2], pooled_features.shape[2],
                                _p(xyz), _p(boxes3d), _p(pts_feature), _p(pooled_features), _p(pooled_empty_flag))
    return 1


def points_in_boxes_cpu(boxes, pts, box_idx_of_points):
    O.lib().orc_points_in_boxes(boxes.shape[0], boxes.shape[1], pts.shape[1], _p(boxes), _p(pts), _p(box_idx_of_points))
    return 1


def sparse_conv3d_dense(features, indices, spatial_shape, batch_size, weight, kernel, stride, padding, subm, cache, key):
    """Oracle of the sparse convolutions (multimodal_gar_amd/sparse_ops.py): the DENSE conv3d of the densified tensor,
    read back at the active output sites (submanifold: the input sites; strided: every output cell whose receptive field
    holds an active input, in ascending (b, z, y, x) order).  Plain torch on CPU, differentiable through autograd."""
    import torch.nn.functional as F
    z, y, x = [int(s) for s in spatial_shape]
    idx = indices.long()
    c = features.shape[1]
    dense = features.new_zeros((batch_size, z, y, x, c))
    dense = dense.index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), features).permute(0, 4, 1, 2, 3)
    w = weight.permute(0, 4, 1, 2, 3)                                  # (Cout, kz, ky, kx, Cin) -> (Cout, Cin, kz, ky, kx)
    st = tuple(stride) if isinstance(stride, (list, tuple)) else (stride,) * 3
    pd = tuple(padding) if isinstance(padding, (list, tuple)) else (padding,) * 3
    kk = tuple(kernel) if isinstance(kernel, (list, tuple)) else (kernel,) * 3
    if subm:
        st, pd = (1, 1, 1), tuple(k // 2 for k in kk)
    out = F.conv3d(dense, w, stride=st, padding=pd)
    if subm:
        oidx = idx
    else:
        occ = features.new_zeros((batch_size, 1, z, y, x))
        occ[idx[:, 0], 0, idx[:, 1], idx[:, 2], idx[:, 3]] = 1.0
        hit = F.conv3d(occ, occ.new_ones((1, 1) + kk), stride=st, padding=pd)[:, 0] > 0.5
        oidx = hit.nonzero()
    feats = out[oidx[:, 0], :, oidx[:, 1], oidx[:, 2], oidx[:, 3]]
    return feats, oidx.int().contiguous(), list(out.shape[2:])


@contextlib.contextmanager
def use_cpu_oracle():
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as bmod
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as smod
    from multimodal_gar_amd import vision_ops, dafm_ops, graph_ops, sparse_ops
    from multimodal_gar_amd.model import gat_model
    O.build()
    saved = []

    def patch(obj, name, new):
        saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, new)

    for cls, mod in ((_Batch, bmod), (_Stack, smod)):
        for name in dir(cls):
            if name.endswith("_wrapper"):
                patch(mod, name, getattr(cls, name))
    patch(vision_ops, "roi_align", roi_align_cpu)
    patch(dafm_ops, "dafm_attention", dafm_attention_cpu)
    patch(gat_model, "dafm_attention", dafm_attention_cpu)
    patch(graph_ops, "_GatAggregate", _GatAggregateCpu)
    patch(sparse_ops, "sparse_conv3d", sparse_conv3d_dense)
    from multimodal_gar_amd.pcdet.ops.roipoint_pool3d import roipoint_pool3d_cuda as rpmod
    from multimodal_gar_amd.pcdet.ops.roiaware_pool3d import roiaware_pool3d_cuda as ramod
    patch(rpmod, "forward", roipoint_pool3d_forward_cpu)
    patch(ramod, "points_in_boxes_gpu", points_in_boxes_cpu)
    try:
        yield
    finally:
        for obj, name, old in reversed(saved):
            setattr(obj, name, old)
