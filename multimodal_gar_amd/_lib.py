"""ctypes binding of libmgar_hip.so -- the ONLY way the Python host code reaches the kernels.

The product path has no CPU fallback: if the HIP library is missing this module raises at
import time, and every op refuses non-device tensors.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmgar_hip.so")
ABI_VERSION = 12

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "multimodal_gar_amd: %s is missing -- build it with `python -m multimodal_gar_amd.build` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)

_cdll = ctypes.CDLL(LIB_PATH)

_I, _F, _P, _LL = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_longlong

# name -> argument ctypes, in the order of include/mgar_ops.h
_PROTOS = {
    "mgar_ball_query_batch": [_I, _I, _I, _F, _I, _P, _P, _P, _P],
    "mgar_group_points_batch": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_group_points_grad_batch": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_gather_points_batch": [_I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_gather_points_grad_batch": [_I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_ball_query_multi_batch": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_ball_query_multi_stack": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_fps_batch": [_I, _I, _I, _P, _P, _P, _P],
    "mgar_morton_codes": [_I, _I, _P, _P, _P],
    "mgar_fps_batch_perm": [_I, _I, _I, _P, _P, _P, _P, _P],
    "mgar_point_grid_workspace_bytes": [_I, ctypes.c_longlong],
    "mgar_point_grid_build": [_I, _I, ctypes.c_longlong, _P, _P, _F, _P, _P],
    "mgar_ball_query_grid_batch": [_I, _I, _I, _F, _I, _P, _P, _P, _P],
    "mgar_ball_query_grid_stack": [_I, _I, ctypes.c_longlong, _F, _I, _P, _P, _P, _P, _P],
    "mgar_three_nn_grid_batch": [_I, _I, _I, _P, _P, _P, _P, _P],
    "mgar_three_nn_grid_stack": [_I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P, _P],
    "mgar_fps_batch_buckets": [_I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_fps_batch_buckets_workspace_floats": [_I, _I],
    "mgar_three_nn_batch": [_I, _I, _I, _P, _P, _P, _P, _P],
    "mgar_three_interpolate_batch": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mgar_three_interpolate_batch_add": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mgar_three_interpolate_grad_batch": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mgar_three_interpolate_grad_sorted_batch": [_I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_three_interpolate_batch_into": [_I, _I, _I, _I, _P, _P, _P, _P, _LL, _P],
    "mgar_three_interpolate_grad_batch_strided": [_I, _I, _I, _I, _P, _LL, _P, _P, _P, _P],
    "mgar_three_interpolate_grad_sorted_batch_strided": [_I, _I, _I, _I, _P, _LL, _P, _P, _P],
    "mgar_ball_query_stack": [_I, _I, _F, _I, _P, _P, _P, _P, _P, _P],
    "mgar_voxel_query_stack": [_I, _I, _I, _I, _I, _F, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_voxel_query_hash_stack": [_I, _I, _I, _I, _I, _F, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P],
    "mgar_fps_stack": [_I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_group_points_stack": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_group_points_grad_stack": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_three_nn_stack": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "mgar_three_interpolate_stack": [_I, _I, _P, _P, _P, _P, _P],
    "mgar_three_interpolate_grad_stack": [_I, _I, _P, _P, _P, _P, _P],
    "mgar_query_group_batch_fwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_query_group_batch_bwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_query_group_stack_fwd": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_query_group_stack_bwd": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_query_group_proj_batch_fwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_query_group_proj_batch_bwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_query_group_proj_stack_fwd": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgar_query_group_proj_stack_bwd": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "mgar_rowmajor_dw_workspace_floats": [_LL, _I, _I],
    "mgar_ktimer_enable": [_I],
    "mgar_ktimer_count": [],
    "mgar_ktimer_add_flops": [_I, ctypes.c_double],
    "mgar_ktimer_add_bytes": [_I, ctypes.c_double],
    "mgar_ktimer_read": [_I, _P, _P, _P, _P, _I],
    "mgar_rowmajor_dw": [_P, _I, _P, _I, _LL, _I, _I, _P, _P, _P],
    "mgar_bn_workspace_floats": [_I, _I, _I],
    "mgar_bn_train_stats": [_P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "mgar_bn_train_stats_grouped": [_P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "mgar_bn_act_fwd_grouped": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    "mgar_bn_act_fwd": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    "mgar_bn_act_maxpool_fwd": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P],
    "mgar_bn_act_bwd": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgar_bn_stats_from_partials_workspace_floats": [_I, _I],
    "mgar_bn_stats_from_partials": [_P, _I, _I, _LL, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "mgar_pointwise_conv_fwd_stats": [_P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P],
    "mgar_query_group_proj_stack_fwd_stats": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "mgar_bn_act_maxpool_bwd_strided": [_P, _LL, _LL, _LL, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgar_bn_cl_workspace_floats": [_I, _I, _I, _I],
    "mgar_bn_cl_train_stats": [_P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "mgar_bn_cl_act_fwd": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P],
    "mgar_bn_rows_bwd_workspace_floats": [_I, _I],
    "mgar_bn_rows_bwd": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgar_bn_act_fwd_to_cl": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P],
    "mgar_maxpool3d_same_fwd_cl": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "mgar_bn_act_small": [_P, _I, _I, _I, _I, _F, _F, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _LL, _P],
    "mgar_bn_act_fwd_into": [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _LL, _P],
    "mgar_bn_act_bwd_rowmajor": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgar_query_group_stack_inverse_items": [_I, _I, _LL],
    "mgar_query_group_stack_inverse_workspace_ints": [_I, _I, _LL],
    "mgar_query_group_stack_inverse_index": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_query_group_stack_bwd_rows": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _LL, _P],
    "mgar_bn_act_maxpool_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P],
    "mgar_pointwise_conv_dw": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "mgar_pointwise_dw_workspace_floats": [_I, _I, _I, _I],
    "mgar_pointwise_dw_bnbwd_workspace_floats": [_I, _I, _I, _I],
    "mgar_pointwise_conv_dw_bnbwd": [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "mgar_bn_act_bwd_apply": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P],
    "mgar_pointwise_conv_dw_act": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P],
    "mgar_pointwise_conv_fwd": [_P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    "mgar_stem_conv3d_workspace_floats": [],
    "mgar_stem_conv3d_set_minimal_filtering": [_I],
    "mgar_stem_conv3d_fwd": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "mgar_delay_us": [_I, _P],
    "mgar_conv3d_k3_workspace_floats": [_I, _I],
    "mgar_conv3d_k3_set_lds_pad": [_I],
    "mgar_conv3d_k3_fwd": [_P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P],
    "mgar_maxpool3d_same_fwd": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "mgar_maxpool3d_valid_fwd": [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "mgar_roi_align_fwd": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _F, _I, _I, _P, _P],
    "mgar_roi_align_bwd": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _F, _I, _I, _P, _P],
    "mgar_dafm_attn_fwd": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P],
    "mgar_dafm_attn_bwd": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "mgar_gatv2_fwd": [_I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P],
    "mgar_gatv2_bwd": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_gatv2_bwd_workspace_floats": [_I, _I, _I, _I],
    "mgar_points_in_boxes": [_I, _I, _I, _P, _P, _P, _P],
    "mgar_roipoint_pool3d_fwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_image_resample_ksize": [_I, _I],
    "mgar_image_resample_coeffs": [_I, _I, _P, _P],
    "mgar_image_resize_normalize_u8": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _LL, _LL, _I, _P],
    "mgar_velodyne_merge_crop_workspace_ints": [_I, _I],
    "mgar_velodyne_merge_crop": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_voxel_hash_build": [_I, _P, _I, _I, _I, _P, _P, _I, _P],
    "mgar_voxel_hash_lookup": [_I, _P, _I, _I, _I, _P, _P, _I, _P, _P],
    "mgar_spconv_rulebook": [_I, _P, _P, _P, _P, _I, _I, _P, _P],
    "mgar_spconv_output_keys": [_I, _P, _P, _P, _P],
    "mgar_spconv_gather_gemm": [_I, _I, _I, _I, _P, _P, _P, _I, _P, _P],
    "mgar_spconv_dw_chunks": [_I],
    "mgar_spconv_pair_chunk": [],
    "mgar_spconv_set_register_gather": [_I],
    "mgar_spconv_pairs_blocks": [_I],
    "mgar_spconv_pairs_count": [_I, _I, _P, _P, _P, _P],
    "mgar_spconv_pairs_fill": [_I, _I, _P, _P, _P, _P, _P, _P],
    "mgar_spconv_pairs_gemm": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_spconv_pairs_dw": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_spconv_dw": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "mgar_voxel_roi_pool_stats_workspace_doubles": [_I, _I],
    "mgar_voxel_roi_pool_bwd_workspace_floats": [_I, _I],
    "mgar_voxel_roi_pool_stats": [_I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_voxel_roi_pool_fwd": [_I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "mgar_voxel_roi_pool_bwd": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P],
}
# bf16-payload twins (include/mgar_ops.h, last section): identical argument lists
for _n in ("mgar_query_group_batch_fwd", "mgar_query_group_stack_fwd", "mgar_query_group_proj_batch_fwd",
           "mgar_query_group_proj_stack_fwd", "mgar_bn_train_stats", "mgar_bn_train_stats_grouped", "mgar_bn_act_fwd",
           "mgar_bn_act_fwd_grouped", "mgar_bn_act_fwd_into", "mgar_bn_act_small", "mgar_bn_cl_train_stats", "mgar_bn_cl_act_fwd", "mgar_bn_act_fwd_to_cl",
           "mgar_maxpool3d_same_fwd_cl", "mgar_bn_act_maxpool_fwd", "mgar_bn_act_bwd", "mgar_bn_act_maxpool_bwd",
           "mgar_pointwise_conv_fwd", "mgar_three_interpolate_batch", "mgar_three_interpolate_batch_into", "mgar_three_interpolate_batch_add", "mgar_three_interpolate_stack",
           "mgar_maxpool3d_same_fwd", "mgar_maxpool3d_valid_fwd", "mgar_roi_align_fwd", "mgar_voxel_roi_pool_fwd", "mgar_stem_conv3d_fwd"):
    _PROTOS[_n + "_bf16"] = _PROTOS[_n]
BF16_TWINS = frozenset(n[:-5] for n in _PROTOS if n.endswith("_bf16"))

# entry points declared `long long` in include/mgar_ops.h (every other one returns int); listed by name, and
# tests/test_capi_cpu.py checks the list against the header's declarations
_LONGLONG_RESULTS = frozenset((
    "mgar_query_group_stack_inverse_items", "mgar_query_group_stack_inverse_workspace_ints", "mgar_bn_cl_workspace_floats",
    "mgar_bn_stats_from_partials_workspace_floats", "mgar_voxel_roi_pool_stats_workspace_doubles",
    "mgar_voxel_roi_pool_bwd_workspace_floats", "mgar_velodyne_merge_crop_workspace_ints", "mgar_gatv2_bwd_workspace_floats",
    "mgar_fps_batch_buckets_workspace_floats", "mgar_point_grid_workspace_bytes",
    "mgar_bn_rows_bwd_workspace_floats", "mgar_conv3d_k3_workspace_floats"))

_fns = {}
for _name, _args in _PROTOS.items():
    _fn = getattr(_cdll, _name)  # AttributeError here = the library is stale: rebuild it
    _fn.argtypes = _args
    _fn.restype = ctypes.c_longlong if _name in _LONGLONG_RESULTS else ctypes.c_int
    _fns[_name] = _fn

_cdll.mgar_abi_version.restype = ctypes.c_int
_cdll.mgar_last_error.restype = ctypes.c_char_p
if _cdll.mgar_abi_version() != ABI_VERSION:
    raise ImportError("libmgar_hip.so ABI %d != expected %d: rebuild" % (_cdll.mgar_abi_version(), ABI_VERSION))


class MgarError(RuntimeError):
    pass


def dev_ptr(t, dtype=None):
    """Raw device pointer of a contiguous device tensor (the C ABI takes nothing else)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MgarError("expected a device (HIP) tensor, got %s -- there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise MgarError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise MgarError("expected dtype %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def fptr(t):
    return dev_ptr(t, torch.float32)


def pptr(t, dtype):
    """Pointer of a feature-payload tensor, which must have the payload dtype of the call (float32 or bfloat16)."""
    return dev_ptr(t, dtype)


def payload_call(name, dtype, *args):
    """`name` for float32 payloads, its `_bf16` twin for bfloat16 ones."""
    if dtype == torch.float32:
        return call(name, *args)
    if dtype == torch.bfloat16 and name in BF16_TWINS:
        return call(name + "_bf16", *args)
    raise MgarError("%s: unsupported payload dtype %s" % (name, dtype))


def iptr(t):
    return dev_ptr(t, torch.int32)


def stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def raw(name, *args):
    """Call an entry point whose return value is not a status code (e.g. a size query)."""
    return _fns[name](*args)


def call(name, *args):
    rc = _fns[name](*args)
    if rc != 0:
        raise MgarError("%s failed with code %d: %s" % (name, rc, _cdll.mgar_last_error().decode()))
    return rc


def host_arrays(radii, nsamples, idx_tensors):
    """ctypes host arrays (float[], int[], void*[]) for the multi-radius ball query entry points."""
    n = len(radii)
    fa = (ctypes.c_float * n)(*[float(r) for r in radii])
    ia = (ctypes.c_int * n)(*[int(s) for s in nsamples])
    pa = (ctypes.c_void_p * n)(*[iptr(t) for t in idx_tensors])
    return fa, ia, pa


def kernel_timers(enable=None, reset=True):
    """enable=True/False switches the per-kernel HIP-event timers of the library; otherwise returns
    {kernel name: (total_ms, launches, algorithmic_bytes, flops)} since the last reset."""
    if enable is not None:
        call("mgar_ktimer_enable", int(bool(enable)))
        _KT_STATE["on"] = bool(enable)
        return None
    _cdll.mgar_ktimer_name.restype = ctypes.c_char_p
    _cdll.mgar_ktimer_name.argtypes = [ctypes.c_int]
    out = {}
    for i in range(raw("mgar_ktimer_count")):
        ms, by, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
        call("mgar_ktimer_read", i, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by), ctypes.byref(fl), int(reset))
        if n.value:
            out[_cdll.mgar_ktimer_name(i).decode()] = (ms.value, n.value, by.value, fl.value)
    return out


_KT_IDS = {}


_KT_STATE = {"on": False}


def note_work(kernel, flops=0.0, nbytes=0.0):
    """Instrumented runs only: credit algorithmic flops / bytes to `kernel` for launches whose work only the caller knows
    (per-sample counts or pair lists that live on the device)."""
    if not _KT_STATE["on"]:
        return
    if not _KT_IDS:
        _cdll.mgar_ktimer_name.restype = ctypes.c_char_p
        _cdll.mgar_ktimer_name.argtypes = [ctypes.c_int]
        for i in range(raw("mgar_ktimer_count")):
            _KT_IDS[_cdll.mgar_ktimer_name(i).decode()] = i
    if flops:
        call("mgar_ktimer_add_flops", _KT_IDS[kernel], float(flops))
    if nbytes:
        call("mgar_ktimer_add_bytes", _KT_IDS[kernel], float(nbytes))


def kernel_timing_on():
    return _KT_STATE["on"]


def note_pair_tests(kernel, pairs):
    """Instrumented runs only: credit `pairs` (query, point) distance evaluations (8 flop each) to `kernel`."""
    note_work(kernel, flops=float(pairs) * 8.0)


def exported_symbols():
    return sorted(_PROTOS) + ["mgar_abi_version", "mgar_last_error", "mgar_ktimer_name"]
