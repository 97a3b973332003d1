"""Drop-in for the reference's compiled module ``pointnet2_stack_cuda``.

Same function names and argument order as the pybind11 table in
pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:12-31 (the four vector-pool
entries are out of scope: PV-RCNN++ only, unreachable from MGAR-net -- SURVEY.md section 2.2).
"""
from ..... import _lib as L


def _note_pairs(kernel, cnt_a, cnt_b, scans=1):
    """Instrumented runs only (bench.py's roofline step): sum_i M_i * N_i pair tests of a stacked-layout scan; the counts
    live on the device, so this syncs -- never on the product path (L.note_pair_tests returns at once when timers are off)."""
    if L._KT_STATE["on"]:
        L.note_pair_tests(kernel, scans * float((cnt_a.double() * cnt_b.double()).sum().item()))


def ball_query_scan_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    """The scan kernel (csrc/ball_query.hip): every (query, point) pair of a sample."""
    L.call("mgar_ball_query_stack", B, M, float(radius), nsample, L.fptr(new_xyz), L.iptr(new_xyz_batch_cnt),
           L.fptr(xyz), L.iptr(xyz_batch_cnt), L.iptr(idx), L.stream_of(xyz))
    _note_pairs("ball_query_kernel", new_xyz_batch_cnt, xyz_batch_cnt)
    return 1


def ball_query_grid_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, grid, idx):
    """The same rows through a uniform cell grid over the clouds (csrc/ball_query_grid.hip); grid: point_grid.PointGrid."""
    L.call("mgar_ball_query_grid_stack", B, M, grid.n_total, float(radius), nsample, L.fptr(new_xyz), L.iptr(new_xyz_batch_cnt),
           L.fptr(grid.ws), L.iptr(idx), L.stream_of(new_xyz))
    return 1


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx, grid=None):
    """grid: a point_grid.PointGrid over (xyz, xyz_batch_cnt) built by the caller (several radii over one cloud set share it);
    without one, large clouds get their own."""
    from ..... import point_grid as G
    if grid is None and B > 0 and M > 0 and xyz.is_cuda and G.wanted(xyz.shape[0] // max(B, 1), [nsample]):
        grid = G.PointGrid(xyz, G.cell_for([radius]), xyz_batch_cnt.int())
    if grid is not None and nsample <= G.MAX_NSAMPLE:
        return ball_query_grid_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, grid, idx)
    return ball_query_scan_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)


def ball_query_multi_wrapper(B, M, radii, nsamples, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_list):
    """Several (radius, nsample) pairs in one scan; idx_list[r] (M, nsamples[r]) as ball_query_wrapper fills it."""
    from ..... import point_grid as G
    if B > 0 and M > 0 and G.wanted(xyz.shape[0] // max(B, 1), nsamples):
        grid = G.PointGrid(xyz, G.cell_for(radii), xyz_batch_cnt.int())
        for r, ns, idx in zip(radii, nsamples, idx_list):
            ball_query_grid_wrapper(B, M, r, ns, new_xyz, new_xyz_batch_cnt, grid, idx)
        return 1
    if len(radii) == 1:
        return ball_query_scan_wrapper(B, M, radii[0], nsamples[0], new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_list[0])
    fa, ia, pa = L.host_arrays(radii, nsamples, idx_list)
    L.call("mgar_ball_query_multi_stack", B, M, len(radii), fa, ia, L.fptr(new_xyz), L.iptr(new_xyz_batch_cnt), L.fptr(xyz),
           L.iptr(xyz_batch_cnt), pa, L.stream_of(xyz))
    _note_pairs("ball_query_kernel", new_xyz_batch_cnt, xyz_batch_cnt)
    return 1


def voxel_query_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                        point_indices, idx):
    L.call("mgar_voxel_query_stack", M, R1, R2, R3, nsample, float(radius), z_range, y_range, x_range,
           L.fptr(new_xyz), L.fptr(xyz), L.iptr(new_coords), L.iptr(point_indices), L.iptr(idx), L.stream_of(xyz))
    return 1


def voxel_query_hash_wrapper(M, R1, R2, R3, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords, table, idx):
    """voxel_query_wrapper with the voxel -> row lookups served by a sparse_ops.VoxelHash instead of the dense table."""
    import torch
    L.call("mgar_voxel_query_hash_stack", M, R1, R2, R3, nsample, float(radius), z_range, y_range, x_range, L.fptr(new_xyz), L.fptr(xyz),
           L.iptr(new_coords), L.dev_ptr(table.keys, torch.int64), L.iptr(table.vals), table.capacity, L.iptr(idx), L.stream_of(xyz))
    return 1


def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    # pointnet2_stack re-exports the dense-batch FPS (pointnet2_stack/src/sampling_gpu.cu:25-140)
    L.call("mgar_fps_batch", b, n, m, L.fptr(points), L.fptr(temp), L.iptr(idx), L.stream_of(points))
    return 1


def stack_farthest_point_sampling_wrapper(points, temp, xyz_batch_cnt, idx, num_sampled_points):
    L.call("mgar_fps_stack", xyz_batch_cnt.shape[0], points.shape[0], L.fptr(points), L.fptr(temp),
           L.iptr(xyz_batch_cnt), L.iptr(idx), L.iptr(num_sampled_points), L.stream_of(points))
    return 1


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    L.call("mgar_group_points_stack", B, M, C, nsample, L.fptr(features), L.iptr(features_batch_cnt), L.iptr(idx),
           L.iptr(idx_batch_cnt), L.fptr(out), L.stream_of(features))
    return 1


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    L.call("mgar_group_points_grad_stack", B, M, C, N, nsample, L.fptr(grad_out), L.iptr(idx),
           L.iptr(idx_batch_cnt), L.iptr(features_batch_cnt), L.fptr(grad_features), L.stream_of(grad_out))
    return 1


def three_nn_grid_wrapper(unknown, unknown_batch_cnt, grid, dist2, idx):
    """three_nn through a cell grid over the known points (csrc/ball_query_grid.hip, three_nn_grid_kernel)."""
    L.call("mgar_three_nn_grid_stack", unknown_batch_cnt.shape[0], unknown.shape[0], grid.n_total, L.fptr(unknown),
           L.iptr(unknown_batch_cnt), L.fptr(grid.ws), L.fptr(dist2), L.iptr(idx), L.stream_of(unknown))


def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    from ..... import point_grid as G
    b = unknown_batch_cnt.shape[0]
    if known.is_cuda and b > 0 and unknown.shape[0] > 0 and G.ENABLED and known.shape[0] // b >= G.MIN_POINTS_PER_CLOUD:
        return three_nn_grid_wrapper(unknown, unknown_batch_cnt.int(), G.PointGrid(known, 0.0, known_batch_cnt.int()), dist2, idx)
    return three_nn_scan_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx)


def three_nn_scan_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    L.call("mgar_three_nn_stack", unknown_batch_cnt.shape[0], unknown.shape[0], known.shape[0], L.fptr(unknown),
           L.iptr(unknown_batch_cnt), L.fptr(known), L.iptr(known_batch_cnt), L.fptr(dist2), L.iptr(idx),
           L.stream_of(unknown))
    _note_pairs("three_nn_kernel", unknown_batch_cnt, known_batch_cnt)


def three_interpolate_wrapper(features, idx, weight, out):
    dt = features.dtype        # feature payload: float32 or bfloat16
    L.payload_call("mgar_three_interpolate_stack", dt, idx.shape[0], features.shape[1], L.pptr(features, dt), L.iptr(idx),
                   L.fptr(weight), L.pptr(out, dt), L.stream_of(features))


def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    L.call("mgar_three_interpolate_grad_stack", idx.shape[0], grad_out.shape[1], L.fptr(grad_out), L.iptr(idx),
           L.fptr(weight), L.fptr(grad_features), L.stream_of(grad_out))


# ---- fused ops that are torch op chains in the reference (no pybind counterpart) ----
def query_group_wrapper(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx_raw, out):
    dt = out.dtype
    L.payload_call("mgar_query_group_stack_fwd", dt, B, M, C, nsample, L.fptr(xyz), L.iptr(xyz_batch_cnt), L.fptr(new_xyz),
                   L.iptr(new_xyz_batch_cnt), L.pptr(features, dt) if features is not None else None, L.iptr(idx_raw),
                   L.pptr(out, dt), L.stream_of(xyz))
    return 1


def query_group_grad_wrapper(B, M, C, nsample, grad_out, idx_raw, new_xyz_batch_cnt, xyz_batch_cnt, grad_features):
    L.call("mgar_query_group_stack_bwd", B, M, C, nsample, L.fptr(grad_out), L.iptr(idx_raw), L.iptr(new_xyz_batch_cnt),
           L.iptr(xyz_batch_cnt), L.fptr(grad_features), L.stream_of(grad_out))
    return 1


def query_group_proj_wrapper(B, M, C, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, zf, wx, idx_raw, rel_out, y_out,
                             zf_ld=None, zf_col=0, out_stats=None):
    """zf: (N, zf_ld) matrix whose columns zf_col .. zf_col + C hold this scale's projection.
    out_stats (C, M * nsample / 128, 2) fp32: also leave the BatchNorm statistics partials of y_out (bn_ops.StatsPartial)."""
    zf_ld = C if zf_ld is None else zf_ld
    dt = zf.dtype
    L.pptr(zf, dt)             # device / contiguity / dtype checks; the call below takes the pointer of the column block
    if out_stats is not None:
        L.call("mgar_query_group_proj_stack_fwd_stats", B, M, C, nsample, L.fptr(xyz), L.iptr(xyz_batch_cnt), L.fptr(new_xyz),
               L.iptr(new_xyz_batch_cnt), zf.data_ptr() + zf.element_size() * zf_col, zf_ld, L.fptr(wx), L.iptr(idx_raw),
               L.fptr(rel_out) if rel_out is not None else None, L.fptr(y_out), L.fptr(out_stats), L.stream_of(xyz))
        return 1
    L.payload_call("mgar_query_group_proj_stack_fwd", dt, B, M, C, nsample, L.fptr(xyz), L.iptr(xyz_batch_cnt), L.fptr(new_xyz),
                   L.iptr(new_xyz_batch_cnt), zf.data_ptr() + zf.element_size() * zf_col, zf_ld, L.fptr(wx), L.iptr(idx_raw),
                   L.pptr(rel_out, dt) if rel_out is not None else None, L.pptr(y_out, dt), L.stream_of(xyz))
    return 1


def query_group_proj_grad_wrapper(B, M, C, nsample, grad_y, idx_raw, new_xyz_batch_cnt, xyz_batch_cnt, grad_zf, zf_ld=None,
                                  zf_col=0):
    zf_ld = C if zf_ld is None else zf_ld
    L.call("mgar_query_group_proj_stack_bwd", B, M, C, nsample, L.fptr(grad_y), L.iptr(idx_raw), L.iptr(new_xyz_batch_cnt),
           L.iptr(xyz_batch_cnt), grad_zf.data_ptr() + 4 * zf_col, zf_ld, L.stream_of(grad_y))
    return 1


def query_group_inverse_index(B, M, nsample, N, idx_raw, new_xyz_batch_cnt, xyz_batch_cnt):
    """Inverted index of a raw ball-query result idx (M, nsample): for every source row the columns (query * nsample + slot)
    that gather it, ascending, cut into work items (csrc/query_group.hip, qg_inv_*_kernel) -> (list, items, row_item, workspace)."""
    import torch
    dev, total = idx_raw.device, M * nsample
    n_items = L.raw("mgar_query_group_stack_inverse_items", B, N, total)
    inv_list = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
    items = torch.zeros((max(n_items, 1), 4), dtype=torch.int32, device=dev)
    row_item = torch.empty((max(N, 1),), dtype=torch.int32, device=dev)
    ws = torch.empty((max(L.raw("mgar_query_group_stack_inverse_workspace_ints", B, N, total), 1),), dtype=torch.int32, device=dev)
    L.call("mgar_query_group_stack_inverse_index", B, M, nsample, N, L.iptr(idx_raw), L.iptr(new_xyz_batch_cnt), L.iptr(xyz_batch_cnt),
           L.iptr(ws), L.iptr(inv_list), L.iptr(items), L.iptr(row_item), L.stream_of(idx_raw))
    return inv_list, items, row_item, ws


def query_group_proj_grad_rows_wrapper(B, M, C, nsample, grad_y_t, idx_raw, new_xyz_batch_cnt, xyz_batch_cnt, grad_zf, zf_ld=None,
                                       zf_col=0, xyz=None, new_xyz=None, index=None):
    """The gradient of query_group_proj_wrapper from a ROW-MAJOR grad_y_t (M * nsample, C): inverted index + one owner per
    source row, no atomics, bit-reproducible (csrc/query_group.hip, qg_stack_bwd_rows_kernel).  grad_zf zero-filled by the
    caller.  With xyz / new_xyz: also returns d wx (C, 3), the gradient of the relative-coordinate weights."""
    import torch
    zf_ld = C if zf_ld is None else zf_ld
    n_rows, total = grad_zf.shape[0], M * nsample
    inv_list, items, row_item, ws = index if index is not None else \
        query_group_inverse_index(B, M, nsample, n_rows, idx_raw, new_xyz_batch_cnt, xyz_batch_cnt)
    n_items = items.shape[0]
    part_rows = torch.empty((n_items, C), dtype=torch.float32, device=grad_zf.device)
    wx_part = torch.zeros(((n_items + 7) // 8, C, 3), dtype=torch.float32, device=grad_zf.device) if xyz is not None else None
    L.call("mgar_query_group_stack_bwd_rows", n_items if total and n_rows else 0, n_rows, C, nsample, L.iptr(ws), L.iptr(items), L.iptr(row_item),
           L.iptr(inv_list), L.fptr(grad_y_t), L.fptr(xyz) if xyz is not None else None, L.fptr(new_xyz) if xyz is not None else None,
           grad_zf.data_ptr() + 4 * zf_col, zf_ld, L.fptr(part_rows), L.fptr(wx_part) if wx_part is not None else None, total,
           L.stream_of(grad_y_t))
    if wx_part is None:
        return None
    return wx_part.sum(0) if total and n_rows else torch.zeros((C, 3), dtype=torch.float32, device=grad_zf.device)


def rowmajor_dw(a, f):
    """a (N, Co), f (N, Ci) row-major contiguous -> a^T f (Co, Ci) on csrc/rowmajor_dw.hip."""
    import torch
    n, co = a.shape
    ci = f.shape[1]
    dw = torch.empty((co, ci), dtype=torch.float32, device=a.device)
    ws = torch.empty((max(1, L.raw("mgar_rowmajor_dw_workspace_floats", n, co, ci)),), dtype=torch.float32, device=a.device)
    L.call("mgar_rowmajor_dw", L.fptr(a), co, L.fptr(f), ci, n, co, ci, L.fptr(ws), L.fptr(dw), L.stream_of(a))
    return dw


# ---- fused Voxel-RoI pooling (csrc/voxel_roi_pool.hip; no pybind counterpart: a torch op chain in the reference) ----
def voxel_roi_pool_stats(M, nsample, C, xyz, new_xyz, idx_raw, w_pos, eps, momentum, moments, mean, invstd, running_mean,
                         running_var, num_batches_tracked):
    import torch
    ws = torch.empty((max(1, L.raw("mgar_voxel_roi_pool_stats_workspace_doubles", M, nsample)),), dtype=torch.float64, device=xyz.device)
    L.call("mgar_voxel_roi_pool_stats", M, nsample, C, L.fptr(xyz), L.fptr(new_xyz), L.iptr(idx_raw), L.fptr(w_pos), float(eps),
           float(momentum), L.dev_ptr(ws, torch.float64), L.dev_ptr(moments, torch.float64), L.fptr(mean), L.fptr(invstd),
           L.fptr(running_mean) if running_mean is not None else None, L.fptr(running_var) if running_var is not None else None,
           L.dev_ptr(num_batches_tracked, torch.int64) if num_batches_tracked is not None else None, L.stream_of(xyz))
    return 1


def voxel_roi_pool_fwd(M, nsample, C, xyz, new_xyz, feats, idx_raw, w_pos, mean, invstd, gamma, beta, pooled, arg):
    import torch
    dt = feats.dtype           # payload of feats / pooled: float32 or bfloat16
    L.payload_call("mgar_voxel_roi_pool_fwd", dt, M, nsample, C, L.fptr(xyz), L.fptr(new_xyz), L.pptr(feats, dt), feats.shape[1],
                   L.iptr(idx_raw), L.fptr(w_pos), L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta), L.pptr(pooled, dt),
                   L.dev_ptr(arg, torch.uint8), L.stream_of(xyz))
    return 1


def voxel_roi_pool_bwd(M, nsample, C, xyz, new_xyz, idx_raw, w_pos, mean, invstd, gamma, moments, train_stats, dpooled, pooled, arg,
                       dfeats, dgamma, dbeta, dw_pos):
    import torch
    ws = torch.empty((max(1, L.raw("mgar_voxel_roi_pool_bwd_workspace_floats", M, C)),), dtype=torch.float32, device=xyz.device)
    L.call("mgar_voxel_roi_pool_bwd", M, nsample, C, L.fptr(xyz), L.fptr(new_xyz), L.iptr(idx_raw), L.fptr(w_pos), L.fptr(mean),
           L.fptr(invstd), L.fptr(gamma), L.dev_ptr(moments, torch.float64) if moments is not None else None, int(train_stats),
           L.fptr(dpooled), L.fptr(pooled), L.dev_ptr(arg, torch.uint8), L.fptr(ws), L.fptr(dfeats) if dfeats is not None else None,
           dfeats.shape[1] if dfeats is not None else C, L.fptr(dgamma), L.fptr(dbeta), L.fptr(dw_pos), L.stream_of(xyz))
    return 1
