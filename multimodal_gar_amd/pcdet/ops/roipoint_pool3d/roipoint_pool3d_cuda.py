"""Drop-in for the reference's compiled module ``roipoint_pool3d_cuda`` (pcdet/ops/roipoint_pool3d/src/roipoint_pool3d.cpp:23-58:
one entry point, ``forward``), on mgar_roipoint_pool3d_fwd of libmgar_hip.so."""
from .... import _lib as L


def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    """xyz (B, N, 3), boxes3d (B, M, 7), pts_feature (B, N, C) -> pooled_features (B, M, S, 3 + C) and pooled_empty_flag (B, M),
    both pre-zeroed by the caller and written in place (roipoint_pool3d.cpp:23-51)."""
    batch_size, pts_num = xyz.shape[0], xyz.shape[1]
    boxes_num, feature_in_len = boxes3d.shape[1], pts_feature.shape[2]
    sampled_pts_num = pooled_features.shape[2]
    L.call("mgar_roipoint_pool3d_fwd", batch_size, pts_num, boxes_num, feature_in_len, sampled_pts_num, L.fptr(xyz), L.fptr(boxes3d),
           L.fptr(pts_feature), L.fptr(pooled_features), L.iptr(pooled_empty_flag), L.stream_of(xyz))
    return 1
