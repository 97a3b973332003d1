"""Drop-in for the one function of the reference's compiled module ``roiaware_pool3d_cuda`` on the hot path's edge:
``points_in_boxes_gpu`` (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:98-116), on mgar_points_in_boxes."""
from .... import _lib as L


def points_in_boxes_gpu(boxes, pts, box_idx_of_points):
    """boxes (B, N, 7), pts (B, P, 3) -> box_idx_of_points (B, P) int32, written in place (-1 = background)."""
    L.call("mgar_points_in_boxes", boxes.shape[0], boxes.shape[1], pts.shape[1], L.fptr(boxes), L.fptr(pts), L.iptr(box_idx_of_points),
           L.stream_of(pts))
    return 1
