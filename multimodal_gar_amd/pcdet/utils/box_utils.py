"""The one helper of the reference's pcdet/utils/box_utils.py the point crop needs (enlarge_box3d, :187-200)."""
from . import common_utils


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """boxes3d (N, 7) [x, y, z, dx, dy, dz, heading]; extra_width [extra_x, extra_y, extra_z] added to the sizes."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    large = boxes3d.clone()
    extra = boxes3d.new_tensor(extra_width)
    large[:, 3:6] += extra.view(1, -1) if extra.dim() > 0 else extra
    return large
