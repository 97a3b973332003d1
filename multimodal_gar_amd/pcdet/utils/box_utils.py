"""The helpers of the reference's pcdet/utils/box_utils.py the point crop and the input pipeline need: boxes_to_corners_3d
(:28-53), mask_boxes_outside_range_numpy (:93-114), enlarge_box3d (:187-200)."""
from . import common_utils


def boxes_to_corners_3d(boxes3d):
    """boxes3d (N, 7) [x, y, z, dx, dy, dz, heading] -> (N, 8, 3) corners: bottom face 0..3 then top face 4..7, each in the
    order (+x +y), (+x -y), (-x -y), (-x +y) of the box frame."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    template = boxes3d.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                                   [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    corners = common_utils.rotate_points_along_z(corners.view(-1, 8, 3), boxes3d[:, 6]).view(-1, 8, 3)
    corners = corners + boxes3d[:, None, 0:3]
    return corners.numpy() if is_numpy else corners


def mask_boxes_outside_range_numpy(boxes, limit_range, min_num_corners=1, use_center_to_filter=True):
    """boxes (N, 7+) numpy -> (N) bool: centre inside the 3-D range, or at least min_num_corners corners inside the x / y range."""
    if boxes.shape[1] > 7:
        boxes = boxes[:, 0:7]
    if use_center_to_filter:
        centers = boxes[:, 0:3]
        return ((centers >= limit_range[0:3]) & (centers <= limit_range[3:6])).all(axis=-1)
    corners = boxes_to_corners_3d(boxes)[:, :, 0:2]
    inside = ((corners >= limit_range[0:2]) & (corners <= limit_range[3:5])).all(axis=2)
    return inside.sum(axis=1) >= min_num_corners


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """boxes3d (N, 7) [x, y, z, dx, dy, dz, heading]; extra_width [extra_x, extra_y, extra_z] added to the sizes."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    large = boxes3d.clone()
    extra = boxes3d.new_tensor(extra_width)
    large[:, 3:6] += extra.view(1, -1) if extra.dim() > 0 else extra
    return large
