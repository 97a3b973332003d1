"""numpy front-end of the C oracle (oracle/mgar_oracle.c) + float64 numpy restatements of
the third-party ops the reference calls (torchvision roi_align / generalized_box_iou,
torch_geometric GATv2Conv, torchmetrics pairwise_*) and of the DAFM attention core.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg may import this module -- as the checker or the timed baseline.
Parity status: "parity unpinned" by reference tests (the reference has none); see
mgar_oracle.c header and DESIGN.md.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmgar_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "mgar_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libmgar_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _cf(x):
    return ctypes.c_float(float(x))


def set_threads(t):
    lib().orc_set_threads(int(t))


def opt_n_threads(n):
    return int(lib().orc_opt_n_threads(int(n)))


# ------------------------------ batch layout ------------------------------
def ball_query_batch(radius, nsample, xyz, new_xyz):
    xyz, px = _f(xyz); new_xyz, pq = _f(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query_batch(b, n, m, _cf(radius), nsample, pq, px, idx.ctypes.data_as(ctypes.c_void_p))
    return idx


def fps_batch(xyz, npoint, temp=None):
    xyz, px = _f(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32) if temp is None else np.ascontiguousarray(temp, np.float32).copy()
    idx = np.zeros((b, npoint), np.int32)
    lib().orc_fps_batch(b, n, npoint, px, temp.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p))
    return idx, temp


def gather_points(points, idx):
    points, pp = _f(points); idx, pi = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().orc_gather_points(b, c, n, m, pp, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx)
    b, c, m = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().orc_gather_points_grad(b, c, n, m, pg, pi, g.ctypes.data_as(ctypes.c_void_p))
    return g


def group_points_batch(points, idx):
    points, pp = _f(points); idx, pi = _i(idx)
    b, c, n = points.shape
    _, npnt, ns = idx.shape
    out = np.zeros((b, c, npnt, ns), np.float32)
    lib().orc_group_points_batch(b, c, n, npnt, ns, pp, pi, out.ctypes.data_as(ctypes.c_void_p))
    return out


def group_points_grad_batch(grad_out, idx, n):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx)
    b, c, npnt, ns = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().orc_group_points_grad_batch(b, c, n, npnt, ns, pg, pi, g.ctypes.data_as(ctypes.c_void_p))
    return g


def three_nn_batch(unknown, known):
    unknown, pu = _f(unknown); known, pk = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((b, n, 3), np.float32); idx = np.zeros((b, n, 3), np.int32)
    lib().orc_three_nn_batch(b, n, m, pu, pk, d2.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p))
    return d2, idx


def three_interpolate_batch(points, idx, weight):
    points, pp = _f(points); idx, pi = _i(idx); weight, pw = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().orc_three_interpolate_batch(b, c, m, n, pp, pi, pw, out.ctypes.data_as(ctypes.c_void_p))
    return out


def three_interpolate_grad_batch(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx); weight, pw = _f(weight)
    b, c, n = grad_out.shape
    g = np.zeros((b, c, m), np.float32)
    lib().orc_three_interpolate_grad_batch(b, c, n, m, pg, pi, pw, g.ctypes.data_as(ctypes.c_void_p))
    return g


# ------------------------------ stack layout ------------------------------
def ball_query_stack(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """Returns the RAW kernel output (idx[row,0] == -1 marks an empty ball)."""
    xyz, px = _f(xyz); new_xyz, pq = _f(new_xyz)
    xc, pxc = _i(xyz_batch_cnt); qc, pqc = _i(new_xyz_batch_cnt)
    M = new_xyz.shape[0]
    idx = np.zeros((M, nsample), np.int32)
    lib().orc_ball_query_stack(len(xc), M, _cf(radius), nsample, pq, pqc, px, pxc, idx.ctypes.data_as(ctypes.c_void_p))
    return idx


def fps_stack(xyz, xyz_batch_cnt, npoints, temp=None):
    xyz, px = _f(xyz); xc, pxc = _i(xyz_batch_cnt); npnt, pn = _i(npoints)
    temp = np.full((xyz.shape[0],), 1e10, np.float32) if temp is None else np.ascontiguousarray(temp, np.float32).copy()
    idx = np.zeros((int(npnt.sum()),), np.int32)
    lib().orc_fps_stack(len(xc), px, temp.ctypes.data_as(ctypes.c_void_p), pxc, idx.ctypes.data_as(ctypes.c_void_p), pn)
    return idx, temp


def group_points_stack(features, features_batch_cnt, idx, idx_batch_cnt):
    features, pf = _f(features); idx, pi = _i(idx)
    fc, pfc = _i(features_batch_cnt); ic, pic = _i(idx_batch_cnt)
    M, ns = idx.shape
    C = features.shape[1]
    out = np.zeros((M, C, ns), np.float32)
    lib().orc_group_points_stack(len(ic), M, C, ns, pf, pfc, pi, pic, out.ctypes.data_as(ctypes.c_void_p))
    return out


def group_points_grad_stack(grad_out, idx, idx_batch_cnt, features_batch_cnt, N):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx)
    fc, pfc = _i(features_batch_cnt); ic, pic = _i(idx_batch_cnt)
    M, C, ns = grad_out.shape
    g = np.zeros((N, C), np.float32)
    lib().orc_group_points_grad_stack(len(ic), M, C, N, ns, pg, pi, pic, pfc, g.ctypes.data_as(ctypes.c_void_p))
    return g


def three_nn_stack(unknown, unknown_batch_cnt, known, known_batch_cnt):
    unknown, pu = _f(unknown); known, pk = _f(known)
    uc, puc = _i(unknown_batch_cnt); kc, pkc = _i(known_batch_cnt)
    N = unknown.shape[0]
    d2 = np.zeros((N, 3), np.float32); idx = np.zeros((N, 3), np.int32)
    lib().orc_three_nn_stack(len(uc), N, pu, puc, pk, pkc, d2.ctypes.data_as(ctypes.c_void_p),
                             idx.ctypes.data_as(ctypes.c_void_p))
    return d2, idx


def three_interpolate_stack(features, idx, weight):
    features, pf = _f(features); idx, pi = _i(idx); weight, pw = _f(weight)
    N, C = idx.shape[0], features.shape[1]
    out = np.zeros((N, C), np.float32)
    lib().orc_three_interpolate_stack(N, C, pf, pi, pw, out.ctypes.data_as(ctypes.c_void_p))
    return out


def three_interpolate_grad_stack(grad_out, idx, weight, M):
    grad_out, pg = _f(grad_out); idx, pi = _i(idx); weight, pw = _f(weight)
    N, C = grad_out.shape
    g = np.zeros((M, C), np.float32)
    lib().orc_three_interpolate_grad_stack(N, C, pg, pi, pw, g.ctypes.data_as(ctypes.c_void_p))
    return g


def voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """RAW kernel output (idx[row,0] == -1 marks an empty neighbourhood)."""
    xyz, px = _f(xyz); new_xyz, pq = _f(new_xyz)
    nc, pnc = _i(new_coords); pin, ppi = _i(point_indices)
    M = nc.shape[0]
    B, R1, R2, R3 = pin.shape
    z_r, y_r, x_r = (int(v) for v in max_range)
    idx = np.zeros((M, nsample), np.int32)
    lib().orc_voxel_query(M, R1, R2, R3, nsample, _cf(radius), z_r, y_r, x_r, pq, px, pnc, ppi,
                          idx.ctypes.data_as(ctypes.c_void_p))
    return idx


def roi_align(inp, rois, output_size, spatial_scale, sampling_ratio=-1, aligned=False):
    inp, pi_ = _f(inp); rois, pr = _f(rois)
    N, C, H, W = inp.shape
    K = rois.shape[0]
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    out = np.zeros((K, C, ph, pw), np.float32)
    lib().orc_roi_align_fwd(pi_, N, C, H, W, pr, K, ph, pw, _cf(spatial_scale), int(sampling_ratio), int(aligned),
                            out.ctypes.data_as(ctypes.c_void_p))
    return out


# ------------------ float64 numpy restatements (third-party arithmetic) ------------------
def generalized_box_iou(a, b):
    """torchvision.ops.generalized_box_iou: IoU - (C - U)/C on xyxy boxes (call sites
    model/gat_model.py:1350,1470,1519)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2]); rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None, :] - inter
    iou = inter / union
    lti = np.minimum(a[:, None, :2], b[None, :, :2]); rbi = np.maximum(a[:, None, 2:], b[None, :, 2:])
    whi = np.clip(rbi - lti, 0, None)
    areai = whi[..., 0] * whi[..., 1]
    return iou - (areai - union) / areai


def pairwise_euclidean_distance(x, zero_diagonal=True):
    """torchmetrics.functional.pairwise_euclidean_distance(x): sqrt(|x|^2+|y|^2-2xy) (call
    sites model/gat_model.py:1471,1520)."""
    x = np.asarray(x, np.float64)
    d = np.sqrt(np.clip((x * x).sum(1)[:, None] + (x * x).sum(1)[None, :] - 2 * x @ x.T, 0, None))
    if zero_diagonal:
        np.fill_diagonal(d, 0)
    return d


def pairwise_cosine_similarity(x, zero_diagonal=False):
    """torchmetrics.functional.pairwise_cosine_similarity (call site model/gat_model.py:1335)."""
    x = np.asarray(x, np.float64)
    xn = x / np.linalg.norm(x, axis=1, keepdims=True)
    s = xn @ xn.T
    if zero_diagonal:
        np.fill_diagonal(s, 0)
    return s


def softmax(x, axis):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def dafm_attention(q, k, v, de, sigma, scale):
    """model/gat_model.py:487-491: E = softmax(-De/sigma, 1); Att = softmax((QK^T * E)*scale, 1); Att V."""
    q, k, v, de = (np.asarray(t, np.float64) for t in (q, k, v, de))
    e = softmax(-(de / sigma), 1)
    att = softmax((q @ k.T) * e * scale, 1)
    return att @ v, att


def gatv2(x, edge_index, w_l, b_l, w_r, b_r, att, bias, heads, out_ch, slope=0.2, concat=False,
          add_self_loops=True):
    """torch_geometric.nn.GATv2Conv forward in eval mode (no dropout), float64.
    x_l = W_l x + b_l, x_r = W_r x + b_r; e_ij = att_h . leaky_relu(x_l[j] + x_r[i]);
    alpha = softmax over incoming edges j -> i (self loops added); out_i = sum_j alpha x_l[j];
    mean over heads when concat=False; + bias.  edge_index[0] = source j, [1] = target i."""
    x = np.asarray(x, np.float64)
    n = x.shape[0]
    ei = np.asarray(edge_index, np.int64)
    if add_self_loops:
        keep = ei[0] != ei[1]
        ei = np.concatenate([ei[:, keep], np.stack([np.arange(n), np.arange(n)])], 1)
    xl = (x @ np.asarray(w_l, np.float64).T + np.asarray(b_l, np.float64)).reshape(n, heads, out_ch)
    xr = (x @ np.asarray(w_r, np.float64).T + np.asarray(b_r, np.float64)).reshape(n, heads, out_ch)
    a = np.asarray(att, np.float64).reshape(heads, out_ch)
    src, dst = ei
    z = xl[src] + xr[dst]
    z = np.where(z > 0, z, slope * z)
    e = (z * a[None]).sum(-1)  # (E, H)
    out = np.zeros((n, heads, out_ch))
    for i in range(n):
        m = dst == i
        if not m.any():
            continue
        al = softmax(e[m], 0)  # (deg, H)
        out[i] = (al[:, :, None] * xl[src[m]]).sum(0)
    out = out.reshape(n, heads * out_ch) if concat else out.mean(1)
    return out + np.asarray(bias, np.float64)


def voxelize_points_loop(points, vsize_xyz, coors_range_xyz, max_num_points_per_voxel, max_num_voxels):
    """The literal sequential algorithm of the voxel generator the reference wraps
    (pcdet/datasets/processor/data_processor.py:15-60 -> spconv Point2VoxelCPU3d, third-party, restated from its published
    behaviour; see multimodal_gar_amd/pcdet/datasets/processor/data_processor.py).  Pure Python loop: small cases only.
    points (N, C) float32 -> (voxels (V, max_points, C), coordinates (V, 3) [z, y, x] int32, num_points (V) int32)."""
    import numpy as np
    points = np.asarray(points, np.float32)
    lo = np.asarray(coors_range_xyz[:3], np.float32)
    vs = np.asarray(vsize_xyz, np.float32)
    grid = np.round((np.asarray(coors_range_xyz[3:6], np.float64) - np.asarray(coors_range_xyz[:3], np.float64))
                    / np.asarray(vsize_xyz, np.float64)).astype(np.int64)
    index, voxels, coords, counts = {}, [], [], []
    for pt in points:
        c = np.floor((pt[:3] - lo) / vs).astype(np.int64)          # float32 arithmetic, as the device path
        if (c < 0).any() or (c >= grid).any():
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        v = index.get(key)
        if v is None:
            if len(voxels) >= max_num_voxels:
                continue
            v = index[key] = len(voxels)
            voxels.append(np.zeros((max_num_points_per_voxel, points.shape[1]), np.float32))
            coords.append(key)
            counts.append(0)
        if counts[v] < max_num_points_per_voxel:
            voxels[v][counts[v]] = pt
            counts[v] += 1
    if not voxels:
        return (np.zeros((0, max_num_points_per_voxel, points.shape[1]), np.float32), np.zeros((0, 3), np.int32), np.zeros((0,), np.int32))
    return np.stack(voxels), np.asarray(coords, np.int32), np.asarray(counts, np.int32)


def points_in_boxes_mask(boxes, pts, margin):
    """(N, P) int32 0/1: box test with an explicit margin (the shape of the reference's points_in_boxes_cpu)."""
    boxes, pb = _f(boxes); pts, pp = _f(pts)
    out = np.zeros((boxes.shape[0], pts.shape[0]), np.int32)
    lib().orc_points_in_boxes_mask(boxes.shape[0], pts.shape[0], pb, pp, _cf(margin), out.ctypes.data_as(ctypes.c_void_p))
    return out


def points_in_boxes(boxes, pts):
    """boxes (B, N, 7), pts (B, P, 3) -> (B, P) int32 index of the first box holding each point, -1 = none."""
    boxes, pb = _f(boxes); pts, pp = _f(pts)
    out = np.full(pts.shape[:2], -1, np.int32)
    lib().orc_points_in_boxes(boxes.shape[0], boxes.shape[1], pts.shape[1], pb, pp, out.ctypes.data_as(ctypes.c_void_p))
    return out


def roipoint_pool3d(xyz, boxes3d, pts_feature, sampled_pts_num):
    """xyz (B, N, 3), boxes3d (B, M, 7) (already enlarged), pts_feature (B, N, C) -> (pooled (B, M, S, 3 + C), empty (B, M))."""
    xyz, px = _f(xyz); boxes3d, pb = _f(boxes3d); pts_feature, pf = _f(pts_feature)
    b, n, _ = xyz.shape
    m, c = boxes3d.shape[1], pts_feature.shape[2]
    pooled = np.zeros((b, m, sampled_pts_num, 3 + c), np.float32)
    empty = np.zeros((b, m), np.int32)
    lib().orc_roipoint_pool3d(b, n, m, c, sampled_pts_num, px, pb, pf, pooled.ctypes.data_as(ctypes.c_void_p),
                              empty.ctypes.data_as(ctypes.c_void_p))
    return pooled, empty


def load_reference_roiaware():
    """The reference's own roiaware_pool3d.cpp built by oracle/build_ref.sh (oracle/_ref/, build container only), or None.
    Opened with lazy binding: the CUDA launchers the file declares are never defined nor called."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "roiaware_pool3d_ref.so")
    if not os.path.exists(path):
        return None
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        import torch  # noqa: F401  (libtorch must be loaded first)
        spec = importlib.util.spec_from_file_location("roiaware_pool3d_ref", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        sys.setdlopenflags(old)


# ---- input pipeline (reference dataloader.py:47-49, 119-131) -----------------------------------
# The reference resizes every stitched JPEG with torchvision ``transforms.Resize`` on a PIL image, i.e. Pillow's
# ``Image.resize(size, BILINEAR)`` (third-party: Pillow, un-vendored; requirements.txt pins it).  Pillow's published
# algorithm (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
# Vertical_8bpc), restated: a separable triangle filter whose support grows with the down-scaling factor; double-precision
# weights normalised per output pixel, rounded to 22-bit fixed point; integer accumulation from 1 << 21, arithmetic shift,
# clamp to a byte; the horizontal pass first, its BYTE result feeding the vertical pass; a pass whose size does not change
# is skipped.  Pinned against Pillow itself in tests/test_input_pipeline_cpu.py (Pillow is in the image, here and on the
# GPU box) and by the committed fixture tests/golden/pil_resize_*.npz.
RESAMPLE_PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size, out_size):
    """-> (bounds (out, 2) int32 [first tap, tap count], kk (out, ksize) int32 fixed-point weights)."""
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                                   # bilinear: support 1
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)                # C's (int) truncates towards zero, as int() does
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0
        for x in range(xmax):                                      # the same left-to-right double sum
            ww += w[x]
        if ww != 0.0:
            w[:xmax] = w[:xmax] / ww
        for x in range(ksize):
            v = w[x] * (1 << RESAMPLE_PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img, out_size):
    """img (L, ...) uint8 -> (out_size, ...) uint8, resampled along axis 0."""
    bounds, kk = resample_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, xmax = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(img.shape[1:], 1 << (RESAMPLE_PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> RESAMPLE_PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bilinear_resize(img, out_h, out_w):
    """img (H, W, C) uint8 -> (out_h, out_w, C) uint8, what ``PIL.Image.fromarray(img).resize((out_w, out_h), BILINEAR)``
    returns."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.shape[1] != out_w:
        img = _resample_axis0(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    if img.shape[0] != out_h:
        img = _resample_axis0(img, out_h)
    return np.ascontiguousarray(img)


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def to_tensor_normalize(img, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """img (H, W, 3) uint8 -> (3, H, W) float32: torchvision ToTensor (v / 255 in float32) then Normalize ((x - mean) / std,
    float32 mean and std, float32 arithmetic) -- dataloader.py:47-49."""
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, np.float32).reshape(3, 1, 1)
    s = np.asarray(std, np.float32).reshape(3, 1, 1)
    return (x - m) / s


def velodyne_merge_crop(upper, lower, tf_upper, tf_lower, point_cloud_range):
    """upper (Nu, C), lower (Nl, C) float32 [x, y, z, features...]; tf_* (3, 4) float32 rigid transforms [R | t] into the base
    frame -> the merged cloud, upper sensor first (dataloader.py:119-128), with the points outside the x / y range removed
    in order (mask_points_by_range, pcdet/utils/common_utils.py:60-63).  xyz' = R @ xyz + t in float32, products summed left
    to right without contraction."""
    out = []
    for pts, tf in ((upper, tf_upper), (lower, tf_lower)):
        pts = np.asarray(pts, np.float32)
        tf = np.asarray(tf, np.float32)
        q = pts.copy()
        for r in range(3):
            q[:, r] = ((tf[r, 0] * pts[:, 0] + tf[r, 1] * pts[:, 1]) + tf[r, 2] * pts[:, 2]) + tf[r, 3]
        out.append(q)
    pc = np.concatenate(out, 0)
    lim = np.asarray(point_cloud_range, np.float32)
    keep = (pc[:, 0] >= lim[0]) & (pc[:, 0] <= lim[3]) & (pc[:, 1] >= lim[1]) & (pc[:, 1] <= lim[4])
    return pc[keep]
