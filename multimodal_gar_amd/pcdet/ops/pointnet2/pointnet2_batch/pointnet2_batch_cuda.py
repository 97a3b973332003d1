"""Drop-in for the reference's compiled module ``pointnet2_batch_cuda``.

Same function names and argument order as the pybind11 table in
pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-24, so the reference's own
``pointnet2_utils.py`` could import this module unchanged.  Each call forwards raw device
pointers to the C ABI of libmgar_hip.so (include/mgar_ops.h); outputs are written in place
into caller-allocated tensors, as in the reference.  Returns 1 like the reference wrappers.
"""
from ..... import _lib as L


def ball_query_scan_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """The scan kernel (csrc/ball_query.hip): every (query, point) pair, early exit once the rows of a wave are full."""
    L.call("mgar_ball_query_batch", b, n, m, float(radius), nsample, L.fptr(new_xyz), L.fptr(xyz), L.iptr(idx),
           L.stream_of(xyz))
    return 1


def ball_query_grid_wrapper(b, n, m, radius, nsample, new_xyz, grid, idx):
    """The same rows through a uniform cell grid over the cloud (csrc/ball_query_grid.hip); grid: point_grid.PointGrid."""
    L.call("mgar_ball_query_grid_batch", b, n, m, float(radius), nsample, L.fptr(new_xyz), L.fptr(grid.ws), L.iptr(idx),
           L.stream_of(new_xyz))
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    from ..... import point_grid as G
    if G.wanted(n, [nsample]):
        return ball_query_grid_wrapper(b, n, m, radius, nsample, new_xyz, G.PointGrid(xyz, G.cell_for([radius])), idx)
    return ball_query_scan_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx)


def ball_query_multi_wrapper(b, n, m, radii, nsamples, new_xyz, xyz, idx_list):
    """Several (radius, nsample) pairs over the same cloud and centres; idx_list[r] (b, m, nsamples[r]) as ball_query_wrapper
    fills it.  Large clouds: one cell grid, one query per radius; otherwise all radii in one scan."""
    from ..... import point_grid as G
    if G.wanted(n, nsamples):
        grid = G.PointGrid(xyz, G.cell_for(radii))
        for r, ns, idx in zip(radii, nsamples, idx_list):
            ball_query_grid_wrapper(b, n, m, r, ns, new_xyz, grid, idx)
        return 1
    if len(radii) == 1:
        return ball_query_scan_wrapper(b, n, m, radii[0], nsamples[0], new_xyz, xyz, idx_list[0])
    fa, ia, pa = L.host_arrays(radii, nsamples, idx_list)
    L.call("mgar_ball_query_multi_batch", b, n, m, len(radii), fa, ia, L.fptr(new_xyz), L.fptr(xyz), pa, L.stream_of(xyz))
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    L.call("mgar_group_points_batch", b, c, n, npoints, nsample, L.fptr(points), L.iptr(idx), L.fptr(out),
           L.stream_of(points))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    L.call("mgar_group_points_grad_batch", b, c, n, npoints, nsample, L.fptr(grad_out), L.iptr(idx),
           L.fptr(grad_points), L.stream_of(grad_out))
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    L.call("mgar_gather_points_batch", b, c, n, npoints, L.fptr(points), L.iptr(idx), L.fptr(out),
           L.stream_of(points))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    L.call("mgar_gather_points_grad_batch", b, c, n, npoints, L.fptr(grad_out), L.iptr(idx), L.fptr(grad_points),
           L.stream_of(grad_out))
    return 1


def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    L.call("mgar_fps_batch", b, n, m, L.fptr(points), L.fptr(temp), L.iptr(idx), L.stream_of(points))
    return 1


# Called (if set) on the launching stream right before the sampling kernel of farthest_point_sampling_pruned_wrapper goes out --
# after its Morton-order preparation.  workload.ClipModel records an event there that the RGB side stream waits for.
BEFORE_SAMPLING_LAUNCH = None


def farthest_point_sampling_pruned_wrapper(b, n, m, points, temp, idx):
    """Same result as farthest_point_sampling_wrapper; points are visited in Morton order so that whole waves can
    skip the distance update of a round (csrc/fps.hip, fps_pruned_kernel).  1024 <= n <= 16384."""
    import torch
    codes = torch.empty((b, n), dtype=torch.int32, device=points.device)
    L.call("mgar_morton_codes", b, n, L.fptr(points), L.iptr(codes), L.stream_of(points))
    perm = torch.sort(codes, dim=1).indices.int()
    if BEFORE_SAMPLING_LAUNCH is not None:
        BEFORE_SAMPLING_LAUNCH()
    L.call("mgar_fps_batch_perm", b, n, m, L.fptr(points), L.fptr(temp), L.iptr(perm), L.iptr(idx), L.stream_of(points))
    return 1


def farthest_point_sampling_buckets_wrapper(b, n, m, points, temp, idx):
    """Same result again for clouds of 16 385 .. 65 536 points (csrc/fps.hip, fps_bucket_kernel): pruning per 256-point unit."""
    import torch
    codes = torch.empty((b, n), dtype=torch.int32, device=points.device)
    L.call("mgar_morton_codes", b, n, L.fptr(points), L.iptr(codes), L.stream_of(points))
    perm = torch.sort(codes, dim=1).indices.int()
    ws = torch.empty((L.raw("mgar_fps_batch_buckets_workspace_floats", b, n),), dtype=torch.float32, device=points.device)
    L.call("mgar_fps_batch_buckets", b, n, m, L.fptr(points), L.fptr(temp), L.iptr(perm), L.fptr(ws), L.iptr(idx), L.stream_of(points))
    return 1


def three_nn_grid_wrapper(b, n, m, unknown, grid, dist2, idx):
    """three_nn through a cell grid over the known points (csrc/ball_query_grid.hip, three_nn_grid_kernel)."""
    L.call("mgar_three_nn_grid_batch", b, n, m, L.fptr(unknown), L.fptr(grid.ws), L.fptr(dist2), L.iptr(idx), L.stream_of(unknown))
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    from ..... import point_grid as G
    if G.ENABLED and m >= G.MIN_POINTS_PER_CLOUD and b > 0 and n > 0:
        return three_nn_grid_wrapper(b, n, m, unknown, G.PointGrid(known, 0.0), dist2, idx)
    return three_nn_scan_wrapper(b, n, m, unknown, known, dist2, idx)


def three_nn_scan_wrapper(b, n, m, unknown, known, dist2, idx):
    L.call("mgar_three_nn_batch", b, n, m, L.fptr(unknown), L.fptr(known), L.fptr(dist2), L.iptr(idx),
           L.stream_of(unknown))
    return 1


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    dt = points.dtype          # feature payload: float32 or bfloat16 (idx int32, weight float32 either way)
    L.payload_call("mgar_three_interpolate_batch", dt, b, c, m, n, L.pptr(points, dt), L.iptr(idx), L.fptr(weight), L.pptr(out, dt),
                   L.stream_of(points))
    return 1


def three_interpolate_add_wrapper(b, c, m, n, points, idx, weight, out):
    """out (b, c, n) += three_interpolate(points, idx, weight)  (csrc/interpolate.hip, accumulate flag)."""
    dt = points.dtype
    L.payload_call("mgar_three_interpolate_batch_add", dt, b, c, m, n, L.pptr(points, dt), L.iptr(idx), L.fptr(weight), L.pptr(out, dt),
                   L.stream_of(points))
    return 1


def three_interpolate_into_wrapper(b, c, m, n, points, idx, weight, merged):
    """three_interpolate into the first c channels of merged (b, c_total, n) (contiguous)."""
    dt = points.dtype
    L.payload_call("mgar_three_interpolate_batch_into", dt, b, c, m, n, L.pptr(points, dt), L.iptr(idx), L.fptr(weight), L.pptr(merged, dt),
                   merged.shape[1] * merged.shape[2], L.stream_of(points))
    return 1


def _sliced_ptr(t, bstride):
    """Pointer of a (b, c, n) float32 device tensor whose samples are `bstride` elements apart (a channel slice of a wider
    contiguous tensor), or of a contiguous one (bstride None)."""
    import torch
    if bstride is None:
        return L.fptr(t), t.shape[1] * t.shape[2]
    assert t.is_cuda and t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) == t.shape[2] and t.stride(0) == bstride
    return t.data_ptr(), bstride


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points, grad_out_bstride=None):
    ptr, bs = _sliced_ptr(grad_out, grad_out_bstride)
    L.call("mgar_three_interpolate_grad_batch_strided", b, c, n, m, ptr, bs, L.iptr(idx), L.fptr(weight),
           L.fptr(grad_points), L.stream_of(grad_out))
    return 1


# ---- fused ops that are torch op chains in the reference (no pybind counterpart) ----
def query_group_wrapper(b, c, n, npoints, nsample, xyz, new_xyz, features, idx, out):
    dt = out.dtype             # payload type of features / out; xyz and new_xyz are float32
    L.payload_call("mgar_query_group_batch_fwd", dt, b, c, n, npoints, nsample, L.fptr(xyz), L.fptr(new_xyz),
                   L.pptr(features, dt) if features is not None else None, L.iptr(idx), L.pptr(out, dt), L.stream_of(xyz))
    return 1


def query_group_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_features):
    L.call("mgar_query_group_batch_bwd", b, c, n, npoints, nsample, L.fptr(grad_out), L.iptr(idx), L.fptr(grad_features),
           L.stream_of(grad_out))
    return 1


def query_group_proj_wrapper(b, c, n, npoints, nsample, xyz, new_xyz, zf, wx, idx, rel_out, y_out):
    dt = zf.dtype              # payload type of zf / rel_out / y_out; xyz, new_xyz, wx are float32
    L.payload_call("mgar_query_group_proj_batch_fwd", dt, b, c, n, npoints, nsample, L.fptr(xyz), L.fptr(new_xyz), L.pptr(zf, dt),
                   L.fptr(wx), L.iptr(idx), L.pptr(rel_out, dt) if rel_out is not None else None, L.pptr(y_out, dt), L.stream_of(xyz))
    return 1


def query_group_proj_grad_wrapper(b, c, n, npoints, nsample, grad_y, idx, grad_zf):
    L.call("mgar_query_group_proj_batch_bwd", b, c, n, npoints, nsample, L.fptr(grad_y), L.iptr(idx), L.fptr(grad_zf),
           L.stream_of(grad_y))
    return 1


def three_interpolate_grad_sorted_wrapper(b, c, n, m, grad_out, entries, grad_points, grad_out_bstride=None):
    ptr, bs = _sliced_ptr(grad_out, grad_out_bstride)
    L.call("mgar_three_interpolate_grad_sorted_batch_strided", b, c, n, m, ptr, bs, L.iptr(entries), L.fptr(grad_points),
           L.stream_of(grad_out))
    return 1
