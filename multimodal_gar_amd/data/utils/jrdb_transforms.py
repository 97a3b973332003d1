"""Sensor -> base-frame transforms of the two JRDB velodynes.

The reference calls ``jt.transform_pts_upper_velodyne_to_base`` / ``jt.transform_pts_lower_velodyne_to_base``
(dataloader.py:125-126) from a module ``data/utils/jrdb_transforms.py`` that is NOT in the reference repository (nor is the
``jrdb_toolkit`` sub-module it ships empty).  PARITY UNPINNED: the numbers below are the defaults of the public JRDB toolkit's
calibration (``calibration/defaults.yaml``: a yaw of 0.085 rad for the upper sensor, none for the lower one, and the two
mounting offsets), restated from the dataset documentation; a site with the original module overrides them with
``set_calibration``.  Points are (3, N), as in the toolkit.
"""
import numpy as np

UPPER_YAW = 0.085
LOWER_YAW = 0.0
UPPER_OFFSET = (-0.019685, 0.0, 1.077382)
LOWER_OFFSET = (-0.019685, 0.0, 0.742092)

_CALIB = {"upper": (UPPER_YAW, UPPER_OFFSET), "lower": (LOWER_YAW, LOWER_OFFSET)}


def set_calibration(sensor, yaw, offset):
    """Override the yaw (rad, about z) and the (x, y, z) offset of ``sensor`` = 'upper' | 'lower'."""
    if sensor not in _CALIB:
        raise KeyError(sensor)
    _CALIB[sensor] = (float(yaw), tuple(float(v) for v in offset))


def rigid_transform(sensor):
    """(3, 4) float32 [R | t] of ``sensor``: R a rotation about z by the yaw, t the mounting offset."""
    yaw, off = _CALIB[sensor]
    c, s = np.cos(yaw), np.sin(yaw)
    tf = np.array([[c, -s, 0.0, off[0]], [s, c, 0.0, off[1]], [0.0, 0.0, 1.0, off[2]]], dtype=np.float64)
    return tf.astype(np.float32)


def _apply(pts, tf):
    pts = np.asarray(pts, dtype=np.float32)
    out = np.empty_like(pts)
    for r in range(3):                                            # float32, left to right: what the device kernel computes
        out[r] = ((tf[r, 0] * pts[0] + tf[r, 1] * pts[1]) + tf[r, 2] * pts[2]) + tf[r, 3]
    return out


def transform_pts_upper_velodyne_to_base(pts):
    """pts (3, N) in the upper velodyne's frame -> (3, N) in the robot base frame."""
    return _apply(pts, rigid_transform("upper"))


def transform_pts_lower_velodyne_to_base(pts):
    """pts (3, N) in the lower velodyne's frame -> (3, N) in the robot base frame."""
    return _apply(pts, rigid_transform("lower"))
