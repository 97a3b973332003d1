"""Replacements for ``torchmetrics.functional.pairwise_cosine_similarity`` and
``pairwise_euclidean_distance`` (model/gat_model.py:8; unpinned third-party dependency, not
installed here).  Restated from the documented definitions; plain torch -- these are (N, N)
matrices with N <= 128 and sit outside the kernels' critical path."""
from typing import Optional

import torch
from torch import Tensor


def _zero_diag(m: Tensor, zero_diagonal: Optional[bool], y_given: bool) -> Tensor:
    if zero_diagonal is None:
        zero_diagonal = not y_given
    if zero_diagonal:
        m = m.clone()
        m.fill_diagonal_(0)
    return m


def pairwise_euclidean_distance(x: Tensor, y: Optional[Tensor] = None, reduction=None,
                                zero_diagonal: Optional[bool] = None) -> Tensor:
    """sqrt(|x|^2 + |y|^2 - 2 x.y), clamped at 0 (torchmetrics computes it in float64 and casts
    back)."""
    yy = x if y is None else y
    xd, yd = x.double(), yy.double()
    d2 = (xd * xd).sum(1, keepdim=True) + (yd * yd).sum(1) - 2 * xd @ yd.T
    dist = d2.clamp(min=0).sqrt().to(x.dtype)
    return _zero_diag(dist, zero_diagonal, y is not None)


def pairwise_cosine_similarity(x: Tensor, y: Optional[Tensor] = None, reduction=None,
                               zero_diagonal: Optional[bool] = None) -> Tensor:
    yy = x if y is None else y
    xn = x / torch.norm(x, p=2, dim=1, keepdim=True)
    yn = yy / torch.norm(yy, p=2, dim=1, keepdim=True)
    return _zero_diag(xn @ yn.T, zero_diagonal, y is not None)
