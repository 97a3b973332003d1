"""CPU tests of the boundary: libmgar_hip.so loads without a GPU and exports every symbol that
include/mgar_ops.h declares (no compute calls here), argument validation returns error codes
instead of exiting, and the host-side index plumbing is correct."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mgar_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mgar_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 25
    for must in ["mgar_ball_query_batch", "mgar_ball_query_stack", "mgar_fps_batch", "mgar_fps_stack",
                 "mgar_voxel_query_stack", "mgar_three_nn_batch", "mgar_group_points_grad_stack",
                 "mgar_roi_align_fwd", "mgar_dafm_attn_bwd", "mgar_gatv2_fwd"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    from multimodal_gar_amd import _lib
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(cdll, name), "libmgar_hip.so does not export %s" % name
    assert sorted(_lib.exported_symbols()) == declared_symbols()
    assert cdll.mgar_abi_version() == _lib.ABI_VERSION


def test_binding_return_types_match_the_header():
    """ctypes' restype per entry point is the header's declared return type (a suffix rule gave
    mgar_pointwise_dw_bnbwd_workspace_floats -- an `int` -- a 64-bit restype)."""
    from multimodal_gar_amd import _lib
    text = open(os.path.join(ROOT, "include", "mgar_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decl = dict((name, ret) for ret, name in re.findall(r"^\s*(long long|int)\s+(mgar_[a-z0-9_]+)\s*\(", text, flags=re.M))
    assert len(decl) > 100
    for name, fn in _lib._fns.items():
        want = ctypes.c_longlong if decl[name] == "long long" else ctypes.c_int
        assert fn.restype is want, name
    assert _lib._LONGLONG_RESULTS == {n for n, r in decl.items() if r == "long long" and n in _lib._fns}


def test_invalid_arguments_return_codes_not_exit():
    """The reference exit(-1)s on bad input (ball_query.cpp:14-26); the C ABI returns a code.
    Argument checks run before any HIP call, so this is safe without a GPU."""
    from multimodal_gar_amd import _lib
    assert _lib._fns["mgar_ball_query_batch"](1, 10, 10, 1.0, 0, None, None, None, None) == -3   # nsample < 1
    assert _lib._fns["mgar_ball_query_batch"](-1, 10, 10, 1.0, 4, None, None, None, None) == -1
    assert _lib._fns["mgar_fps_batch"](2, 0, 5, None, None, None, None) == -1                    # m > 0 from empty
    assert _lib._fns["mgar_three_nn_batch"](0, 0, 0, None, None, None, None, None) == 0          # empty is a no-op
    assert _lib._fns["mgar_gatv2_fwd"](4, 8, 100, None, None, None, None, None, 0.2, None, None, None, None) in (-1, -3)
    with pytest.raises(_lib.MgarError):
        _lib.call("mgar_group_points_batch", -1, 1, 1, 1, 1, None, None, None, None)
    assert b"negative" in _lib._cdll.mgar_last_error()


def test_ops_refuse_cpu_tensors():
    from multimodal_gar_amd import _lib
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as pb
    with pytest.raises(_lib.MgarError):
        pb.ball_query(1.0, 4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))


def test_edges_to_csr_matches_reference_graph_construction():
    from multimodal_gar_amd.graph_ops import edges_to_csr
    n = 5
    comb = torch.combinations(torch.arange(n), r=2)
    ei = torch.cat((comb, torch.flip(comb, [1])), 0).T        # model/gat_model.py:1085-1092
    rowptr, col = edges_to_csr(ei, n)
    assert rowptr.tolist() == [0, 5, 10, 15, 20, 25]
    for i in range(n):
        assert sorted(col[rowptr[i]:rowptr[i + 1]].tolist()) == list(range(n))  # all sources + self loop


def test_scene_offsets_and_roi_format():
    from multimodal_gar_amd.dafm_ops import scene_offsets
    from multimodal_gar_amd.vision_ops import convert_boxes_to_roi_format
    so, do = scene_offsets([3, 1, 4], "cpu")
    assert so.tolist() == [0, 3, 4, 8] and do.tolist() == [0, 9, 10]
    rois = convert_boxes_to_roi_format([torch.ones(2, 4), torch.zeros(1, 4)])
    assert rois.shape == (3, 5) and rois[:, 0].tolist() == [0.0, 0.0, 1.0]
