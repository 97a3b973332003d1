"""Replacements for the torchvision.ops functions the reference calls on the hot path
(``import torchvision.ops as TO``, model/gat_model.py:4): roi_align (HIP kernel) and
generalized_box_iou (plain torch).  torchvision is not installed in this image and is a
third-party dependency of the reference (requirements.txt:422, torchvision==0.17.2); the
arithmetic is restated from its documented semantics."""
from typing import List, Union

import torch
from torch import Tensor
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L


def convert_boxes_to_roi_format(boxes: List[Tensor]) -> Tensor:
    """list of (L_i, 4) xyxy -> (sum L_i, 5) [batch_index, x1, y1, x2, y2]."""
    cat = torch.cat(boxes, dim=0)
    ids = torch.cat([torch.full_like(b[:, :1], i) for i, b in enumerate(boxes)], dim=0)
    return torch.cat([ids, cat], dim=1)


class _RoIAlign(Function):
    @staticmethod
    def forward(ctx, inp, rois, spatial_scale, ph, pw, sampling_ratio, aligned):
        inp = inp.contiguous()
        if inp.dtype != torch.bfloat16:
            inp = inp.float()
        dt = inp.dtype                        # feature payload: float32 or bfloat16; boxes stay float32
        rois = rois.contiguous().float()
        n, c, h, w = inp.shape
        k = rois.shape[0]
        out = torch.empty((k, c, ph, pw), dtype=dt, device=inp.device)
        L.payload_call("mgar_roi_align_fwd", dt, L.pptr(inp, dt), n, c, h, w, L.fptr(rois), k, ph, pw, float(spatial_scale),
                       int(sampling_ratio), int(bool(aligned)), L.pptr(out, dt), L.stream_of(inp))
        ctx.save_for_backward(rois)
        ctx.cfg = (inp.shape, float(spatial_scale), ph, pw, int(sampling_ratio), int(bool(aligned)))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        (n, c, h, w), scale, ph, pw, sr, al = ctx.cfg
        grad_in = None
        if ctx.needs_input_grad[0]:
            grad_in = torch.zeros((n, c, h, w), dtype=torch.float32, device=grad_out.device)
            grad_out = grad_out.contiguous().float()
            L.call("mgar_roi_align_bwd", L.fptr(grad_out), n, c, h, w, L.fptr(rois), rois.shape[0], ph, pw,
                   scale, sr, al, L.fptr(grad_in), L.stream_of(grad_out))
        return grad_in, None, None, None, None, None, None


def roi_align(input: Tensor, boxes: Union[Tensor, List[Tensor]], output_size, spatial_scale: float = 1.0,
              sampling_ratio: int = -1, aligned: bool = False) -> Tensor:
    """Same signature and semantics as torchvision.ops.roi_align."""
    rois = boxes if isinstance(boxes, Tensor) else convert_boxes_to_roi_format(boxes)
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    return _RoIAlign.apply(input, rois, spatial_scale, ph, pw, sampling_ratio, aligned)


def box_area(boxes: Tensor) -> Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def generalized_box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """GIoU = IoU - (C - U) / C for xyxy boxes, (N, M).  Degenerate (zero-area) boxes give
    NaN exactly as torchvision's 0/0 does; the reference guards against that upstream
    (train_func.py:103-109)."""
    a1, a2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2 - inter
    iou = inter / union
    lti = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rbi = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    whi = (rbi - lti).clamp(min=0)
    areai = whi[..., 0] * whi[..., 1]
    return iou - (areai - union) / areai
