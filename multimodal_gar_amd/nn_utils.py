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


class _PointwiseConv(torch.autograd.Function):
    """y = W x for x (B, Cin, P): forward and dX are library GEMMs; the weight gradient -- a GEMM with
    a tiny (Cout x Cin) output and up to 15.7 M reduction columns -- runs on csrc/pointwise_dw.hip."""

    @staticmethod
    def forward(ctx, x3, w):
        ctx.save_for_backward(x3, w)
        return torch.bmm(w.unsqueeze(0).expand(x3.shape[0], -1, -1), x3)

    @staticmethod
    def backward(ctx, dy):
        from . import _lib as L
        x3, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.bmm(w.t().unsqueeze(0).expand(dy.shape[0], -1, -1), dy)
        if ctx.needs_input_grad[1]:
            dw = pointwise_dw(x3, dy)
        return dx, dw


_DW_MIN_COLUMNS = 1 << 16   # below this the library GEMM is fine


def pointwise_dw(x3, dy):
    """sum_b dy[b] @ x3[b]^T -> (Cout, Cin) on csrc/pointwise_dw.hip (x3 (B, Cin, P), dy (B, Cout, P))."""
    from . import _lib as L
    x3, dy = x3.contiguous(), dy.contiguous()
    b, cin, p = x3.shape
    cout = dy.shape[1]
    dw = torch.empty((cout, cin), dtype=torch.float32, device=dy.device)
    ws = torch.empty((max(1, L.raw("mgar_pointwise_dw_workspace_floats", b, cin, cout, p)),), dtype=torch.float32, device=dy.device)
    L.call("mgar_pointwise_conv_dw", L.fptr(x3), L.fptr(dy), b, cin, cout, p, L.fptr(ws), L.fptr(dw), L.stream_of(dy))
    return dw


def conv1x1(conv, x):
    """y[b, o, ...] = sum_i W[o, i] x[b, i, ...] (+ bias) for a kernel-size-1 ConvNd."""
    w = conv.weight.view(1, conv.out_channels, conv.in_channels)
    x3 = x.flatten(2)
    # 3-D @ 3-D: the result is produced directly as (B, C_out, P).  (A 2-D weight makes torch fold
    # the batch into the GEMM's rows and hand back a TRANSPOSED view, which the next op then
    # materialises with a slow strided copy of the whole activation.)
    if (x3.is_cuda and x3.dtype == torch.float32 and x3.shape[0] * x3.shape[2] >= _DW_MIN_COLUMNS
            and conv.out_channels <= 64 and conv.in_channels <= 128   # one pass over both operands (OB <= 2, IB <= 4)
            and torch.is_grad_enabled() and conv.weight.requires_grad):
        y = _PointwiseConv.apply(x3, w[0].contiguous())
    else:
        y = torch.bmm(w.expand(x3.shape[0], -1, -1), x3)
    if conv.bias is not None:
        y = y + conv.bias.view(1, -1, 1)
    return y.view(x.shape[0], conv.out_channels, *x.shape[2:])


def _is_pointwise(m):
    return isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)) and all(k == 1 for k in m.kernel_size) \
        and all(s == 1 for s in m.stride) and all(p == 0 for p in m.padding) and m.groups == 1


_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


class PointwiseSequential(nn.Sequential):
    """nn.Sequential (same state-dict keys) for the shared MLPs: kernel-size-1 convolutions run as
    GEMMs, and on the device every BatchNorm [+ ReLU] that follows runs in the fused HIP kernels of
    csrc/bn_act.hip; ``forward_maxpool`` additionally folds the max over the last axis (nsample)
    into the last BatchNorm + ReLU so that activation is never written."""

    fuse_bn_conv = True

    def _run(self, x, pool_last, start=0, rowmajor_input_grad=False, input_stats=None):
        from . import bn_ops
        layers = list(self)
        i, n = start, len(layers)
        pooled = False
        stats = input_stats          # BatchNorm statistics partials of the current x, if its producer kernel left them
        while i < n:
            layer = layers[i]
            # only the layer that consumes the input can hand its gradient back row-major
            rm = rowmajor_input_grad and i == start and x.dim() == 4
            if _is_pointwise(layer):
                # a conv with no BatchNorm in front (the module's first layer): the streaming kernel also leaves the statistics
                # partials of its output for the BatchNorm that follows
                feeds_bn = i + 1 < n and isinstance(layers[i + 1], _BN_TYPES) and layers[i + 1].training
                y = bn_ops.plain_conv(x, layer, want_out_stats=feeds_bn) if x.is_cuda and self.fuse_bn_conv else None
                if y is not None:
                    x, stats = y if feeds_bn else (y, None)
                else:
                    x, stats = conv1x1(layer, x), None
                i += 1
            elif isinstance(layer, _BN_TYPES) and x.is_cuda:
                relu = i + 1 < n and isinstance(layers[i + 1], nn.ReLU)
                last = i + (2 if relu else 1) >= n
                y = None
                nxt = i + (2 if relu else 1)
                if nxt < n and _is_pointwise(layers[nxt]) and self.fuse_bn_conv:
                    # [BN -> ReLU -> next conv] in one pass; the activated tensor is never written.  If a BatchNorm follows that
                    # conv, the kernel also leaves the statistics partials of its output (no statistics pass over it)
                    feeds_bn = nxt + 1 < n and isinstance(layers[nxt + 1], _BN_TYPES) and layers[nxt + 1].training
                    y = bn_ops.bn_act_conv(x, layer, relu, layers[nxt], rowmajor_grad=rm, in_stats=stats, want_out_stats=feeds_bn)
                    if y is not None:
                        x, stats = y if feeds_bn else (y, None)
                        i = nxt + 1
                        continue
                if pool_last and last and x.dim() == 4:
                    y = bn_ops.bn_act_maxpool(x, layer, relu, in_stats=stats)
                    pooled = y is not None
                if y is None:
                    y = bn_ops.bn_act(x, layer, relu, rowmajor_grad=rm, in_stats=stats)
                if y is None:            # eval-mode backward etc.: plain torch
                    y = layer(x)
                    relu = False
                x, stats = y, None
                i += 2 if relu else 1
            else:
                x = layer(x)
                stats = None
                i += 1
        return x, pooled

    def forward(self, x):
        return self._run(x, False)[0]

    def forward_maxpool(self, x, start=0, rowmajor_input_grad=False, input_stats=None):
        """(B, C, M, ns) -> (B, C', M): the MLP (from layer `start`) followed by a max over ns.
        ``rowmajor_input_grad``: the caller's backward prefers d x as (M * ns, C) rows (B = 1; see bn_ops._bwd_rowmajor)."""
        y, pooled = self._run(x, True, start, rowmajor_input_grad, input_stats)
        return y if pooled else y.max(dim=3).values

    def first_layer_foldable(self, c_in):
        """True if folding the first layer into the grouping shrinks the gathered tensor: a bias-free
        point-wise conv whose output is narrower than its (3 + C) input."""
        conv = self[0] if len(self) else None
        return conv is not None and _is_pointwise(conv) and conv.bias is None and conv.in_channels == c_in \
            and conv.out_channels < c_in
