"""Dense helpers shared by the host-side modules.

Point-wise (kernel-size-1) convolutions are evaluated as GEMMs on the parameters of the
``nn.ConvNd`` module they belong to (parameter names / shapes stay the reference's).  On ROCm a 1x1
``Conv2d`` over a (120, 4..64, 4096, 32) grouped tensor goes to MIOpen, whose immediate-mode
fallback picks naive direct kernels for these tall, narrow shapes (minutes per backward at config
c3); the same contraction as a strided-batched GEMM goes to hipBLASLt / rocBLAS and runs at
memory speed.  Arithmetic is identical up to fp32 summation order.
"""
import torch
import torch.nn as nn


def conv1x1(conv, x):
    """y[b, o, ...] = sum_i W[o, i] x[b, i, ...] (+ bias) for a kernel-size-1 ConvNd."""
    w = conv.weight.view(conv.out_channels, conv.in_channels)
    y = torch.matmul(w, x.flatten(2))
    if conv.bias is not None:
        y = y + conv.bias.view(1, -1, 1)
    return y.view(x.shape[0], conv.out_channels, *x.shape[2:])


def _is_pointwise(m):
    return isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)) and all(k == 1 for k in m.kernel_size) \
        and all(s == 1 for s in m.stride) and all(p == 0 for p in m.padding) and m.groups == 1


class PointwiseSequential(nn.Sequential):
    """nn.Sequential (same state-dict keys) whose kernel-size-1 convolutions run as GEMMs."""

    def forward(self, x):
        for layer in self:
            x = conv1x1(layer, x) if _is_pointwise(layer) else layer(x)
        return x
