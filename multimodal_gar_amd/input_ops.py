"""Per-clip input preparation on the device: csrc/input_prep.hip behind the C ABI.

* ``resize_normalize`` -- what the reference's ``transforms.Compose([Resize(image_size), ToTensor(), Normalize(mean, std)])``
  (dataloader.py:47-49) does to every decoded frame, for all frames of a clip in one launch, from uint8 frames in HBM:
  Pillow's bilinear resample bit for bit, then ((v / 255) - mean) / std exactly as torch evaluates it in float32.
* ``resize_bytes`` -- the resampled bytes alone (the stage tests compare with ``PIL.Image.resize``).
* ``velodyne_merge_crop`` -- dataloader.py:119-128 (``load_pc``) + the range mask of ``mask_points_and_boxes_outside_range``.

No CPU path: the tensors must live on the GPU and the HIP library must be present.
"""
import ctypes

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)        # dataloader.py:49
IMAGENET_STD = (0.229, 0.224, 0.225)

_TABLES = {}
_LUTS = {}


def resample_tables_host(in_size, out_size):
    """Pillow's per-axis coefficient tables from the library's host helper: bounds (out, 2) int32, kk (out, ksize) int32."""
    ksize = _lib.raw("mgar_image_resample_ksize", int(in_size), int(out_size))
    if ksize <= 0:
        raise _lib.MgarError("mgar_image_resample_ksize(%d, %d) failed" % (in_size, out_size))
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    _lib.call("mgar_image_resample_coeffs", int(in_size), int(out_size), bounds.ctypes.data_as(ctypes.c_void_p),
              kk.ctypes.data_as(ctypes.c_void_p))
    return bounds, kk


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _TABLES:
        b, k = resample_tables_host(in_size, out_size)
        _TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device))
    return _TABLES[key]


def normalize_lut(mean, std, device):
    """(3, 256) float32: ((v / 255) - mean[c]) / std[c] in float32, the way ToTensor + Normalize compute it."""
    key = (tuple(float(m) for m in mean), tuple(float(s) for s in std), str(device))
    if key not in _LUTS:
        v = torch.arange(256, dtype=torch.float32).div(255).view(1, 256)
        m = torch.as_tensor(key[0], dtype=torch.float32).view(3, 1)
        s = torch.as_tensor(key[1], dtype=torch.float32).view(3, 1)
        _LUTS[key] = v.sub(m).div(s).contiguous().to(device)
    return _LUTS[key]


def _launch(frames, size, lut, dst, fs, cs, kind):
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise _lib.MgarError("frames must be (F, H, W, 3) uint8")
    f, ih, iw, _ = frames.shape
    oh, ow = int(size[0]), int(size[1])
    xb, xk = _tables(iw, ow, frames.device)
    yb, yk = _tables(ih, oh, frames.device)
    _lib.call("mgar_image_resize_normalize_u8", f, ih, iw, oh, ow, _lib.dev_ptr(frames, torch.uint8), _lib.iptr(xb), _lib.iptr(xk),
              _lib.iptr(yb), _lib.iptr(yk), None if lut is None else _lib.fptr(lut), dst.data_ptr(), fs, cs, kind,
              _lib.stream_of(frames))


def resize_bytes(frames, size):
    """frames (F, H, W, 3) uint8 on the GPU -> (F, size[0], size[1], 3) uint8: ``PIL.Image.resize((w, h), BILINEAR)`` per frame."""
    out = torch.empty((frames.shape[0], int(size[0]), int(size[1]), 3), dtype=torch.uint8, device=frames.device)
    _launch(frames, size, None, out, 0, 0, 2)
    return out


def resize_normalize(frames, size, mean=IMAGENET_MEAN, std=IMAGENET_STD, dtype=torch.float32, layout="tchw", out=None):
    """frames (F, H, W, 3) uint8 on the GPU -> normalised planes of ``size`` = (h, w):
    layout "tchw": (F, 3, h, w), one clip as the reference's loader returns it (dataloader.py:273);
    layout "cthw": (3, F, h, w), the clip as the I3D trunk reads it (no permute pass afterwards).
    ``out``: a contiguous tensor of that shape / dtype to write into (e.g. one clip of the batch tensor)."""
    f, oh, ow = frames.shape[0], int(size[0]), int(size[1])
    if layout not in ("tchw", "cthw"):
        raise ValueError("layout is 'tchw' or 'cthw'")
    if dtype not in (torch.float32, torch.bfloat16):
        raise _lib.MgarError("resize_normalize: dtype is float32 or bfloat16")
    shape = (f, 3, oh, ow) if layout == "tchw" else (3, f, oh, ow)
    if out is None:
        out = torch.empty(shape, dtype=dtype, device=frames.device)
    elif tuple(out.shape) != shape or out.dtype != dtype or not out.is_contiguous() or out.device != frames.device:
        raise _lib.MgarError("resize_normalize: `out` must be a contiguous %s tensor of shape %s on the frames' device" % (dtype, shape))
    plane = oh * ow
    fs, cs = (3 * plane, plane) if layout == "tchw" else (plane, f * plane)
    _launch(frames, size, normalize_lut(mean, std, frames.device), out, fs, cs, 0 if dtype == torch.float32 else 1)
    return out


def velodyne_merge_crop(upper, lower, tf_upper, tf_lower, point_cloud_range):
    """upper (Nu, C), lower (Nl, C) float32 on the GPU, tf_* (3, 4) [R | t] (host values), point_cloud_range the 6 numbers of
    the config -> (K, C): both clouds in the base frame, upper first, the points outside the x / y range dropped, order kept."""
    if upper.dim() != 2 or lower.dim() != 2 or upper.shape[1] != lower.shape[1] or upper.shape[1] < 3:
        raise _lib.MgarError("velodyne_merge_crop: clouds must be (N, C >= 3) with the same C")
    nu, nl, c = upper.shape[0], lower.shape[0], upper.shape[1]
    tu = (ctypes.c_float * 12)(*np.asarray(tf_upper, np.float32).reshape(12).tolist())
    tl = (ctypes.c_float * 12)(*np.asarray(tf_lower, np.float32).reshape(12).tolist())
    r = np.asarray(point_cloud_range, np.float32)
    xy = (ctypes.c_float * 4)(float(r[0]), float(r[1]), float(r[3]), float(r[4]))
    dev = upper.device
    ws = torch.empty((_lib.raw("mgar_velodyne_merge_crop_workspace_ints", nu, nl),), dtype=torch.int32, device=dev)
    out = torch.empty((nu + nl, c), dtype=torch.float32, device=dev)
    count = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.call("mgar_velodyne_merge_crop", nu, nl, c, _lib.fptr(upper) if nu else None, _lib.fptr(lower) if nl else None,
              ctypes.cast(tu, ctypes.c_void_p), ctypes.cast(tl, ctypes.c_void_p), ctypes.cast(xy, ctypes.c_void_p),
              _lib.iptr(ws), _lib.fptr(out) if nu + nl else None, _lib.iptr(count), _lib.stream_of(upper))
    return out[:int(count.item())]
