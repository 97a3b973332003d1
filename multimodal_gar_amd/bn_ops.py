"""Fused BatchNorm(training statistics) + ReLU (+ max over nsample) on the HIP kernels of
csrc/bn_act.hip -- the tail of every [Conv 1x1 -> BatchNorm -> ReLU] block of the shared MLPs
(reference pointnet2_batch/pointnet2_modules.py:37-45 and friends).  Parameters and running
statistics live in the caller's ``nn.BatchNormNd`` module, so state dicts are unchanged."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L


def _u8ptr(t):
    return L.dev_ptr(t, torch.uint8)


def _workspace(x, b, c, p):
    return torch.empty((L.raw("mgar_bn_workspace_floats", b, c, p),), dtype=torch.float32, device=x.device)


_PAYLOADS = (torch.float32, torch.bfloat16)   # feature-payload dtypes of the kernels; statistics / affine are always fp32


def _f32(t):
    """BatchNorm vectors (affine, statistics) are fp32 whatever the payload type (a module converted with .to(bf16)
    would otherwise hand bf16 vectors to kernels that read floats)."""
    return t if t is None or t.dtype == torch.float32 else t.float()


STATS_TILE = 128   # columns per statistics partial of the producer kernels (csrc/pointwise_fwd.hip, csrc/query_group.hip)


def stats_partial_buffer(x_like, c, n_per_channel):
    """Buffer a producer kernel fills with the BatchNorm statistics partials of its output -- (C, n / 128, 2): per channel
    and 128-column tile the tile's mean and sum of squared deviations -- or None where that does not apply (n % 128, dtype)."""
    if x_like.dtype != torch.float32 or not x_like.is_cuda or n_per_channel % STATS_TILE != 0 or n_per_channel == 0:
        return None
    return torch.empty((c, n_per_channel // STATS_TILE, 2), dtype=torch.float32, device=x_like.device)


def _train_stats(x3, bn, partial=None):
    b, c, p = x3.shape
    mean = torch.empty((c,), dtype=torch.float32, device=x3.device)
    invstd = torch.empty_like(mean)
    track = bn.track_running_stats and bn.running_mean is not None
    if partial is not None and tuple(partial.shape) == (c, (b * p) // STATS_TILE, 2) and (b * p) % STATS_TILE == 0:
        # the producer of x left the per-tile (mean, M2): merge them, no pass over x (csrc/bn_act.hip, bn_finalize_kernel)
        nws = L.raw("mgar_bn_stats_from_partials_workspace_floats", partial.shape[1], c)
        ws = torch.empty((nws,), dtype=torch.float32, device=x3.device) if nws else None
        L.call("mgar_bn_stats_from_partials", L.fptr(partial), partial.shape[1], c, b * p, STATS_TILE, float(bn.eps),
               float(bn.momentum if bn.momentum is not None else 0.1), L.fptr(ws) if ws is not None else None, L.fptr(mean), L.fptr(invstd),
               L.fptr(bn.running_mean) if track else None, L.fptr(bn.running_var) if track else None,
               L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None,
               L.stream_of(x3))
        return mean, invstd
    ws = _workspace(x3, b, c, p)
    L.payload_call("mgar_bn_train_stats", x3.dtype, L.pptr(x3, x3.dtype), b, c, p, float(bn.eps),
                   float(bn.momentum if bn.momentum is not None else 0.1),
                   L.fptr(ws), L.fptr(mean), L.fptr(invstd), L.fptr(bn.running_mean) if track else None,
                   L.fptr(bn.running_var) if track else None,
                   L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None,
                   L.stream_of(x3))
    return mean, invstd


ROWMAJOR_GRAD_MAX_CHANNELS = 64   # csrc/bn_act.hip: bn_bwd_apply_t_kernel transposes through a 64 x 65 LDS tile


def _bwd_rowmajor(dy, x3, mean, invstd, gamma, beta, relu, ws, dgamma, dbeta):
    """BatchNorm [+ ReLU] backward whose input gradient is stored ROW-MAJOR, (B * P, C), and handed to autograd as the
    (B, C, P) transposed VIEW of that memory: the consumer that asked for it (the atomic-free stacked grouping backward,
    pointnet2_stack/pointnet2_utils.py) reads whole rows; anything else sees an ordinary strided tensor."""
    b, c, p = x3.shape
    dx_t = torch.empty((b, p, c), dtype=torch.float32, device=x3.device)
    L.call("mgar_bn_act_bwd_rowmajor", L.fptr(dy), L.fptr(x3), b, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
           int(relu), L.fptr(ws), L.fptr(dgamma), L.fptr(dbeta), L.fptr(dx_t), L.stream_of(x3))
    return dx_t.permute(0, 2, 1)


class _BnAct(Function):
    """y = [relu](batch_norm_train(x)) for x (B, C, P); saves x (not y) for the backward."""

    @staticmethod
    def forward(ctx, x3, gamma, beta, mean, invstd, relu, rowmajor_grad=False):
        b, c, p = x3.shape
        ctx.rowmajor = bool(rowmajor_grad) and x3.dtype == torch.float32 and c <= ROWMAJOR_GRAD_MAX_CHANNELS
        y = torch.empty_like(x3)
        dt = x3.dtype
        L.payload_call("mgar_bn_act_fwd", dt, L.pptr(x3, dt), b, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
                       int(relu), L.pptr(y, dt), L.stream_of(x3))
        ctx.save_for_backward(x3, gamma, beta, mean, invstd)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x3, gamma, beta, mean, invstd = ctx.saved_tensors
        b, c, p = x3.shape
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        # every tensor whose pointer is passed stays bound to a name until the call returns: a
        # temporary freed mid-expression can be handed out again by the caching allocator
        dt = x3.dtype
        dy_c, ws = dy.contiguous().to(dt), _workspace(x3, b, c, p)
        if ctx.rowmajor:
            return _bwd_rowmajor(dy_c, x3, mean, invstd, gamma, beta, ctx.relu, ws, dgamma, dbeta), dgamma, dbeta, None, None, None, None
        dx = torch.empty_like(x3)
        L.payload_call("mgar_bn_act_bwd", dt, L.pptr(dy_c, dt), L.pptr(x3, dt), b, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma),
                       L.fptr(beta), int(ctx.relu), L.fptr(ws), L.fptr(dgamma), L.fptr(dbeta), L.pptr(dx, dt), L.stream_of(x3))
        return dx, dgamma, dbeta, None, None, None, None


class _BnActMaxPool(Function):
    """pooled (B, C, M) = max_s [relu](batch_norm_train(x))[b, c, m, s] for x (B, C, M, ns)."""

    @staticmethod
    def forward(ctx, x4, gamma, beta, mean, invstd, relu):
        b, c, m, ns = x4.shape
        dt = x4.dtype
        out = torch.empty((b, c, m), dtype=dt, device=x4.device)
        arg = torch.empty((b, c, m), dtype=torch.uint8, device=x4.device)
        xarg = torch.empty((b, c, m), dtype=dt, device=x4.device) if x4.requires_grad or gamma.requires_grad else None
        L.payload_call("mgar_bn_act_maxpool_fwd", dt, L.pptr(x4, dt), b, c, m, ns, L.fptr(mean), L.fptr(invstd), L.fptr(gamma),
                       L.fptr(beta), int(relu), L.pptr(out, dt), _u8ptr(arg), L.pptr(xarg, dt) if xarg is not None else None,
                       L.stream_of(x4))
        ctx.save_for_backward(x4, gamma, mean, invstd, out, arg, xarg)
        ctx.relu = bool(relu)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    @once_differentiable
    def backward(ctx, dpool, darg=None):
        x4, gamma, mean, invstd, out, arg, xarg = ctx.saved_tensors
        b, c, m, ns = x4.shape
        dx = torch.empty_like(x4)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        dt = x4.dtype
        ws = _workspace(x4, b, c, m * ns)
        if dt == torch.float32 and dpool.dtype == dt and dpool.dim() == 3 and not dpool.is_contiguous() \
                and (b == 1 or dpool.stride(0) >= 0) and min(dpool.stride(1), dpool.stride(2)) >= 1:
            # a channel slice of the concatenated scales, or the transposed view of (M, C_total) rows: read in place
            L.call("mgar_bn_act_maxpool_bwd_strided", dpool.data_ptr(), dpool.stride(0) if b > 1 else 0, dpool.stride(1), dpool.stride(2),
                   L.fptr(out), _u8ptr(arg), L.fptr(x4), L.fptr(xarg) if xarg is not None else None, b, c, m, ns, L.fptr(mean),
                   L.fptr(invstd), L.fptr(gamma), int(ctx.relu), L.fptr(ws), L.fptr(dgamma), L.fptr(dbeta), L.fptr(dx), L.stream_of(x4))
            return dx, dgamma, dbeta, None, None, None
        dpool_c = dpool.contiguous().to(dt)
        L.payload_call("mgar_bn_act_maxpool_bwd", dt, L.pptr(dpool_c, dt), L.pptr(out, dt), _u8ptr(arg), L.pptr(x4, dt),
                       L.pptr(xarg, dt) if xarg is not None else None, b, c, m, ns, L.fptr(mean), L.fptr(invstd), L.fptr(gamma),
                       int(ctx.relu), L.fptr(ws), L.fptr(dgamma), L.fptr(dbeta), L.pptr(dx, dt), L.stream_of(x4))
        return dx, dgamma, dbeta, None, None, None


class _BnActConv(Function):
    """y = W . [relu](batch_norm_train(x)) for x (B, C, P), W (Cout, C): one [BN -> ReLU -> Conv 1x1] step of
    a shared MLP on csrc/pointwise_fwd.hip.  The activated tensor is never written: the forward
    applies BN + ReLU to x while feeding the MFMA, the backward recomputes it inside the weight-
    gradient kernel (csrc/pointwise_dw.hip); only the pre-BN x is saved."""

    @staticmethod
    def forward(ctx, x3, gamma, beta, mean, invstd, relu, w, rowmajor_grad=False, out_stats=None):
        """out_stats (optional, stats_partial_buffer(y)): filled with the statistics partials of y for the BatchNorm that follows."""
        b, c, p = x3.shape
        ctx.rowmajor = bool(rowmajor_grad) and x3.dtype == torch.float32 and c <= ROWMAJOR_GRAD_MAX_CHANNELS
        cout = w.shape[0]
        w = w.contiguous()
        dt = x3.dtype
        y = torch.empty((b, cout, p), dtype=dt, device=x3.device)
        if out_stats is not None:
            L.call("mgar_pointwise_conv_fwd_stats", L.fptr(x3), b, c, p, L.fptr(w), c, 1, cout, L.fptr(mean), L.fptr(invstd),
                   L.fptr(gamma), L.fptr(beta), int(relu), L.fptr(y), L.fptr(out_stats), L.stream_of(x3))
        else:
            L.payload_call("mgar_pointwise_conv_fwd", dt, L.pptr(x3, dt), b, c, p, L.fptr(w), c, 1, cout, L.fptr(mean), L.fptr(invstd),
                           L.fptr(gamma), L.fptr(beta), int(relu), L.pptr(y, dt), L.stream_of(x3))
        ctx.save_for_backward(x3, gamma, beta, mean, invstd, w)
        ctx.relu = relu
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x3, gamma, beta, mean, invstd, w = ctx.saved_tensors
        b, c, p = x3.shape
        cout = w.shape[0]
        gy = gy.contiguous()
        st = L.stream_of(x3)
        # grad wrt the activated input: W^T gy (the forward kernel, W read transposed, no activation)
        ga = torch.empty_like(x3)
        L.call("mgar_pointwise_conv_fwd", L.fptr(gy), b, cout, p, L.fptr(w), 1, c, c, None, None, None, None, 0,
               L.fptr(ga), st)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        if FUSED_DW_BN_REDUCE and ctx.needs_input_grad[6]:
            # dW and the BatchNorm backward's reduction {sum dz, sum dz xhat} from ONE pass over (x, gy) on the matrix cores
            # (csrc/pointwise_dw.hip, pair mode): no reduction pass over (W^T gy, x); then the apply pass alone
            dw = torch.empty_like(w)
            coef = torch.empty((2 * c,), dtype=torch.float32, device=x3.device)
            wsd = torch.empty((max(1, L.raw("mgar_pointwise_dw_bnbwd_workspace_floats", b, c, cout, p)),), dtype=torch.float32,
                              device=x3.device)
            L.call("mgar_pointwise_conv_dw_bnbwd", L.fptr(x3), L.fptr(gy), L.fptr(w), b, c, cout, p, L.fptr(mean), L.fptr(invstd),
                   L.fptr(gamma), L.fptr(beta), int(ctx.relu), L.fptr(wsd), L.fptr(dw), L.fptr(dgamma), L.fptr(dbeta), L.fptr(coef), st)
            dx = torch.empty((b, p, c) if ctx.rowmajor else (b, c, p), dtype=torch.float32, device=x3.device)
            L.call("mgar_bn_act_bwd_apply", L.fptr(ga), L.fptr(x3), b, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
                   int(ctx.relu), L.fptr(coef), int(ctx.rowmajor), L.fptr(dx), st)
            return (dx.permute(0, 2, 1) if ctx.rowmajor else dx), dgamma, dbeta, None, None, None, dw, None, None
        dw = None
        if ctx.needs_input_grad[6]:
            dw = torch.empty_like(w)
            wsd = torch.empty((max(1, L.raw("mgar_pointwise_dw_workspace_floats", b, c, cout, p)),), dtype=torch.float32,
                              device=x3.device)
            L.call("mgar_pointwise_conv_dw_act", L.fptr(x3), L.fptr(gy), b, c, cout, p, L.fptr(mean), L.fptr(invstd),
                   L.fptr(gamma), L.fptr(beta), int(ctx.relu), L.fptr(wsd), L.fptr(dw), st)
        ws = _workspace(x3, b, c, p)
        if ctx.rowmajor:
            return _bwd_rowmajor(ga, x3, mean, invstd, gamma, beta, ctx.relu, ws, dgamma, dbeta), dgamma, dbeta, None, None, None, dw, None, None
        dx = torch.empty_like(x3)
        L.call("mgar_bn_act_bwd", L.fptr(ga), L.fptr(x3), b, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma),
               L.fptr(beta), int(ctx.relu), L.fptr(ws), L.fptr(dgamma), L.fptr(dbeta), L.fptr(dx), st)
        return dx, dgamma, dbeta, None, None, None, dw, None, None


class _PlainConv(Function):
    """y = W x for the FIRST layer of a shared MLP (no BatchNorm in front of it): csrc/pointwise_fwd.hip with the identity in
    place of the activation, so the output is written once at streaming rate and -- when a BatchNorm follows -- the kernel
    leaves that BatchNorm's statistics partials behind (no statistics pass over the widest tensor of the module)."""

    @staticmethod
    def forward(ctx, x3, w, out_stats=None):
        b, c, p = x3.shape
        cout = w.shape[0]
        w = w.contiguous()
        dt = x3.dtype
        y = torch.empty((b, cout, p), dtype=dt, device=x3.device)
        if out_stats is not None:
            L.call("mgar_pointwise_conv_fwd_stats", L.fptr(x3), b, c, p, L.fptr(w), c, 1, cout, None, None, None, None, 0,
                   L.fptr(y), L.fptr(out_stats), L.stream_of(x3))
        else:
            L.payload_call("mgar_pointwise_conv_fwd", dt, L.pptr(x3, dt), b, c, p, L.fptr(w), c, 1, cout, None, None, None, None, 0,
                           L.pptr(y, dt), L.stream_of(x3))
        ctx.save_for_backward(x3, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x3, w = ctx.saved_tensors
        b, c, p = x3.shape
        cout = w.shape[0]
        gy = gy.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x3)
            L.call("mgar_pointwise_conv_fwd", L.fptr(gy), b, cout, p, L.fptr(w), 1, c, c, None, None, None, None, 0, L.fptr(dx),
                   L.stream_of(x3))
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            wsd = torch.empty((max(1, L.raw("mgar_pointwise_dw_workspace_floats", b, c, cout, p)),), dtype=torch.float32,
                              device=x3.device)
            L.call("mgar_pointwise_conv_dw", L.fptr(x3), L.fptr(gy), b, c, cout, p, L.fptr(wsd), L.fptr(dw), L.stream_of(x3))
        return dx, dw, None


PLAIN_CONV_MAX_OUT = 32      # one 32-row output block (csrc/pointwise_fwd.hip OB = 1): above it the library GEMM is as fast
PLAIN_CONV_MAX_IN = 32


def plain_conv(x, conv, want_out_stats=False):
    """conv(x) for a bias-free kernel-size-1 ``conv`` that has no BatchNorm in front (the first layer of a shared MLP), or None
    outside the kernel's shapes (the caller takes the library GEMM).  -> y, or (y, statistics partials | None)."""
    if not (x.is_cuda and x.dtype in _PAYLOADS and x.dim() >= 3 and conv.bias is None):
        return None
    if x.dtype != torch.float32 and torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return None
    c, cout = x.shape[1], conv.out_channels
    x3 = x.contiguous().flatten(2)
    if c > PLAIN_CONV_MAX_IN or cout > PLAIN_CONV_MAX_OUT or x3.shape[2] % 4 != 0 or x3.shape[0] * x3.shape[2] < (1 << 16):
        return None
    out_stats = stats_partial_buffer(x3, cout, x3.shape[0] * x3.shape[2]) if want_out_stats and x3.shape[2] % STATS_TILE == 0 else None
    y = _PlainConv.apply(x3, _f32(conv.weight).view(cout, c), out_stats).view(x.shape[0], cout, *x.shape[2:])
    return (y, out_stats) if want_out_stats else y


# _BnActConv.backward: dW and the BatchNorm-backward reduction from one pass (csrc/pointwise_dw.hip, pair mode) instead of dW,
# then reduce + apply.  Measured at c3 (round 2): bn_bwd_partial 7.3 -> 3.4 ms per step, but pointwise_dw 6.5 -> 12.6 ms -- the
# dW kernel is bound by its LDS staging and operand reads, not by HBM, so doubling its (virtual) input channels doubles its
# time: 228.7 -> 232.5 ms per step.  Kept (tested, tests/test_fused_stats_gpu.py) but off.
FUSED_DW_BN_REDUCE = False
FUSED_CONV_MAX_CHANNELS = 64   # csrc/pointwise_fwd.hip: Cout <= 64 in both directions


FUSED_STATS_MAX_CHANNELS = 32   # csrc/pointwise_fwd.hip: the statistics epilogue exists for Cout <= 32


def bn_act_conv(x, bn, relu, conv, rowmajor_grad=False, in_stats=None, want_out_stats=False):
    """conv(relu?(bn(x))) for a bias-free kernel-size-1 ``conv`` in one fused step, or None if the
    shapes are outside the fused kernel (the caller then runs bn_act and the GEMM separately)."""
    if not (x.is_cuda and x.dtype in _PAYLOADS and bn.training and x.dim() >= 3 and conv.bias is None):
        return None
    if x.dtype != torch.float32 and torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return None   # the weight-gradient kernel (csrc/pointwise_dw.hip) takes fp32 payloads only: bf16 is a forward path
    c, cout = x.shape[1], conv.out_channels
    x3 = x.contiguous().flatten(2)
    if c > FUSED_CONV_MAX_CHANNELS or cout > FUSED_CONV_MAX_CHANNELS or x3.shape[2] % 4 != 0 \
            or x3.shape[0] * x3.shape[2] < (1 << 16):
        return None
    mean, invstd = _stats(x3, bn, in_stats)
    gamma, beta = _affine(bn, c, x.device)
    out_stats = stats_partial_buffer(x3, cout, x3.shape[0] * x3.shape[2]) \
        if want_out_stats and cout <= FUSED_STATS_MAX_CHANNELS and x3.shape[2] % STATS_TILE == 0 else None
    y = _BnActConv.apply(x3, gamma, beta, mean, invstd, relu, _f32(conv.weight).view(cout, c), rowmajor_grad, out_stats)
    y = y.view(x.shape[0], cout, *x.shape[2:])
    return (y, out_stats) if want_out_stats else y


def _affine(bn, c, device):
    if bn.affine:
        return _f32(bn.weight), _f32(bn.bias)
    return torch.ones(c, device=device), torch.zeros(c, device=device)


def _stats(x3, bn, partial=None):
    if bn.training or not bn.track_running_stats:
        return _train_stats(x3, bn, partial)
    return _f32(bn.running_mean), torch.rsqrt(_f32(bn.running_var) + bn.eps)


def _slice_stride(out, x):
    """out: a (B, C, ...) CHANNEL SLICE of a wider contiguous (B, C_total, ...) tensor (what y[:, c0:c1] is) with x's shape
    and dtype -> its batch stride in elements, or None if it is anything else."""
    if out is None or out.shape != x.shape or out.dtype != x.dtype or not out.is_cuda:
        return None
    inner = 1
    for size, stride in zip(reversed(out.shape[1:]), reversed(out.stride()[1:])):
        if size != 1 and stride != inner:
            return None
        inner *= size
    return out.stride(0) if out.stride(0) >= inner else None


# ---- channels-last (NDHWC) activations of the frozen I3D (csrc/channels_last.hpp) -------------------------------------------
def is_channels_last_3d(x):
    return x.dim() == 5 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last_3d)


def _cl_out_slice(out, x):
    """out = y[:, c0:c0 + C] of a channels-last (N, C_total, T, H, W) tensor with x's shape / dtype -> its row stride C_total."""
    if out is None or out.shape != x.shape or out.dtype != x.dtype or not out.is_cuda:
        return None
    n, c, t, h, w = out.shape
    ld = out.stride(4)
    ok = out.stride(1) == 1 and ld >= c and out.stride(3) == w * ld and out.stride(2) == h * w * ld and out.stride(0) == t * h * w * ld
    return ld if ok and ld % 4 == 0 and (out.data_ptr() // out.element_size()) % 4 == 0 else None


def _cl_stats(x, bn, per_sample):
    n, c, t, h, w = x.shape
    r = t * h * w
    rows = n * c if per_sample else c
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    ws = torch.empty((max(L.raw("mgar_bn_cl_workspace_floats", n, r, c, int(per_sample)), 1),), dtype=torch.float32, device=x.device)
    track = bn.track_running_stats and bn.running_mean is not None
    L.payload_call("mgar_bn_cl_train_stats", x.dtype, x.data_ptr(), n, r, c, int(per_sample), float(bn.eps),
                   float(bn.momentum if bn.momentum is not None else 0.1), L.fptr(ws), L.fptr(mean), L.fptr(invstd),
                   L.fptr(bn.running_mean) if track else None, L.fptr(bn.running_var) if track else None,
                   L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None,
                   L.stream_of(x))
    return mean, invstd


def bn_act_channels_last(x, bn, relu, per_sample, out=None):
    """[relu](bn_train(x)) for a channels-last x (forward only; frozen I3D); ``out``: a channel slice of a wider channels-last
    tensor (the Inception concatenation).  None if the shapes do not qualify (C % 4, C > 1024, dtype)."""
    n, c, t, h, w = x.shape
    if c % 4 != 0 or c > 1024 or x.dtype not in _PAYLOADS or (per_sample and n * c > 65535) or not bn.training:
        return None
    mean, invstd = _cl_stats(x, bn, per_sample)
    gamma, beta = _affine(bn, c, x.device)
    ld = _cl_out_slice(out, x)
    y = out if ld is not None else torch.empty_like(x)           # preserve_format: channels-last
    L.payload_call("mgar_bn_cl_act_fwd", x.dtype, x.data_ptr(), n, t * h * w, c, int(per_sample), L.fptr(mean), L.fptr(invstd),
                   L.fptr(gamma), L.fptr(beta), int(relu), y.data_ptr(), ld if ld is not None else c, L.stream_of(x))
    return y


def bn_act_to_channels_last(x, bn, relu, per_sample):
    """The same for a contiguous NCDHW x, RESULT channels-last: where the I3D's activations change layout (after the stem)."""
    x = x.contiguous()
    n, c, t, h, w = x.shape
    if x.dtype not in _PAYLOADS or not bn.training or (per_sample and n * c > 65535):
        return None
    x3 = x.flatten(2)
    if per_sample:
        mean = torch.empty((n * c,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        track = bn.track_running_stats and bn.running_mean is not None
        ws = _workspace(x3, 1, n * c, x3.shape[2])
        L.payload_call("mgar_bn_train_stats_grouped", x.dtype, L.pptr(x3, x.dtype), n, c, x3.shape[2], float(bn.eps),
                       float(bn.momentum if bn.momentum is not None else 0.1), L.fptr(ws), L.fptr(mean), L.fptr(invstd),
                       L.fptr(bn.running_mean) if track else None, L.fptr(bn.running_var) if track else None,
                       L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None,
                       L.stream_of(x))
    else:
        mean, invstd = _train_stats(x3, bn)
    gamma, beta = _affine(bn, c, x.device)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last_3d)
    L.payload_call("mgar_bn_act_fwd_to_cl", x.dtype, L.pptr(x3, x.dtype), n, c, t * h * w, int(per_sample), L.fptr(mean), L.fptr(invstd),
                   L.fptr(gamma), L.fptr(beta), int(relu), y.data_ptr(), c, L.stream_of(x))
    return y


class _BnActRows(Function):
    """y = [relu](batch_norm_train(x)) for ROW-MAJOR x (rows, C): the sparse trunk's features (csrc/channels_last.hpp)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, invstd, relu):
        rows, c = x.shape
        y = torch.empty_like(x)
        L.call("mgar_bn_cl_act_fwd", L.fptr(x), 1, rows, c, 0, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta), int(relu),
               L.fptr(y), c, L.stream_of(x))
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        rows, c = x.shape
        dy = dy.contiguous()
        dx, dgamma, dbeta = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty((L.raw("mgar_bn_rows_bwd_workspace_floats", rows, c),), dtype=torch.float32, device=x.device)
        L.call("mgar_bn_rows_bwd", L.fptr(dy), L.fptr(x), rows, c, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
               int(ctx.relu), L.fptr(ws), L.fptr(dgamma), L.fptr(dbeta), L.fptr(dx), L.stream_of(x))
        return dx, dgamma, dbeta, None, None, None


def bn_act_rows(x, bn, relu):
    """[relu](bn(x)) for a row-major (rows, C) fp32 device tensor and an nn.BatchNorm1d in TRAIN mode (batch statistics, running
    statistics updated), forward + backward on the row-major kernels; None if the shapes do not qualify (C % 4, C > 1024, eval)."""
    if not (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and bn.training and x.shape[1] % 4 == 0 and x.shape[1] <= 1024
            and x.shape[0] > 0):
        return None
    x = x.contiguous()
    rows, c = x.shape
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    ws = torch.empty((max(L.raw("mgar_bn_cl_workspace_floats", 1, rows, c, 0), 1),), dtype=torch.float32, device=x.device)
    track = bn.track_running_stats and bn.running_mean is not None
    L.call("mgar_bn_cl_train_stats", L.fptr(x), 1, rows, c, 0, float(bn.eps), float(bn.momentum if bn.momentum is not None else 0.1),
           L.fptr(ws), L.fptr(mean), L.fptr(invstd), L.fptr(bn.running_mean) if track else None, L.fptr(bn.running_var) if track else None,
           L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None, L.stream_of(x))
    gamma, beta = _affine(bn, c, x.device)
    return _BnActRows.apply(x, gamma, beta, mean, invstd, relu)


SMALL_CHANNEL_MAX = 16384   # csrc/bn_act.hip, bn_small_fused_kernel: elements per channel one workgroup keeps in registers


def _small_fused(x, x3, bn, relu, per_sample, out):
    """Statistics + apply in one launch for small channels (forward only), or None if the shapes do not qualify."""
    b, c, p = x3.shape
    if (p if per_sample else b * p) > SMALL_CHANNEL_MAX or p % 4 != 0 or x3.dtype not in _PAYLOADS:
        return None
    bstride = _slice_stride(out, x)
    y = out if bstride is not None else torch.empty_like(x)
    track = bn.track_running_stats and bn.running_mean is not None
    rows = b * c if per_sample else c
    need = per_sample and track
    mean = torch.empty((rows,), dtype=torch.float32, device=x.device) if need else None
    ws = torch.empty((rows,), dtype=torch.float32, device=x.device) if need else None
    gamma, beta = _affine(bn, c, x.device)
    dt = x3.dtype
    L.payload_call("mgar_bn_act_small", dt, L.pptr(x3, dt), b, c, p, int(per_sample), float(bn.eps),
                   float(bn.momentum if bn.momentum is not None else 0.1), L.fptr(gamma), L.fptr(beta), int(relu),
                   L.fptr(ws) if ws is not None else None, L.fptr(mean) if mean is not None else None, None,
                   L.fptr(bn.running_mean) if track else None, L.fptr(bn.running_var) if track else None,
                   L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None,
                   y.data_ptr(), bstride if bstride is not None else -1, L.stream_of(x3))
    return y


def _apply_into(x3, bn, relu, mean, invstd, per_sample, out, bstride):
    b, c, p = x3.shape
    gamma, beta = _affine(bn, c, x3.device)
    dt = x3.dtype
    L.payload_call("mgar_bn_act_fwd_into", dt, L.pptr(x3, dt), b, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
                   int(relu), int(per_sample), out.data_ptr(), bstride, L.stream_of(x3))
    return out


def bn_act(x, bn, relu, rowmajor_grad=False, out=None, in_stats=None, to_channels_last=False):
    """[relu](bn(x)) for x (B, C, ...) contiguous on the device; ``bn`` is the nn.BatchNormNd module.
    ``out``: write the result into this channel slice of a wider tensor (forward-only callers; see _slice_stride).
    A channels-last 5-D x (forward only) stays channels-last; ``to_channels_last``: NCDHW in, channels-last out."""
    if x.is_cuda and x.dim() == 5 and not (torch.is_grad_enabled() and (x.requires_grad or (bn.affine and bn.weight.requires_grad))):
        y = bn_act_channels_last(x, bn, relu, False, out) if is_channels_last_3d(x) else \
            (bn_act_to_channels_last(x, bn, relu, False) if to_channels_last else None)
        if y is not None:
            return y
    x3 = x.contiguous().flatten(2) if x.dim() > 2 else x.contiguous().unsqueeze(-1)
    if not bn.training and torch.is_grad_enabled() and x.requires_grad:
        return None  # eval-mode backward: let the caller take the plain torch path
    forward_only = not (torch.is_grad_enabled() and (x.requires_grad or (bn.affine and bn.weight.requires_grad)))
    if forward_only and bn.training and x.is_cuda and in_stats is None:
        y = _small_fused(x, x3, bn, relu, False, out)
        if y is not None:
            return y
    mean, invstd = _stats(x3, bn, in_stats)
    bstride = _slice_stride(out, x)
    if bstride is not None and x.dtype in _PAYLOADS and not (torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad)):
        return _apply_into(x3, bn, relu, mean, invstd, False, out, bstride)
    gamma, beta = _affine(bn, x3.shape[1], x.device)
    return _BnAct.apply(x3, gamma, beta, mean, invstd, relu, rowmajor_grad).view(x.shape)


def bn_act_per_sample(x, bn, relu, out=None, to_channels_last=False):
    """[relu](bn(x)) where every sample of x (G, C, ...) is normalised with its own batch statistics and the running
    statistics get the G momentum updates in sample order: G clips through a train-mode BatchNorm in ONE pass, with
    the result of feeding them one at a time.  Forward only (frozen I3D); None if that does not apply."""
    if not (x.is_cuda and x.dtype in _PAYLOADS and bn.training and x.dim() >= 3) or (torch.is_grad_enabled() and
                                                                                        (x.requires_grad or bn.weight.requires_grad)):
        return None
    if x.dim() == 5:
        y = bn_act_channels_last(x, bn, relu, True, out) if is_channels_last_3d(x) else \
            (bn_act_to_channels_last(x, bn, relu, True) if to_channels_last else None)
        if y is not None:
            return y
    x3 = x.contiguous().flatten(2)
    g, c, p = x3.shape
    if g * c > 65535:
        return None
    y = _small_fused(x, x3, bn, relu, True, out)
    if y is not None:
        return y
    mean = torch.empty((g * c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    track = bn.track_running_stats and bn.running_mean is not None
    ws = _workspace(x3, 1, g * c, p)
    st = L.stream_of(x3)
    dt = x3.dtype
    L.payload_call("mgar_bn_train_stats_grouped", dt, L.pptr(x3, dt), g, c, p, float(bn.eps),
                   float(bn.momentum if bn.momentum is not None else 0.1),
                   L.fptr(ws), L.fptr(mean), L.fptr(invstd), L.fptr(bn.running_mean) if track else None,
                   L.fptr(bn.running_var) if track else None,
                   L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None, st)
    bstride = _slice_stride(out, x)
    if bstride is not None:
        return _apply_into(x3, bn, relu, mean, invstd, True, out, bstride)
    gamma, beta = _affine(bn, c, x.device)
    y = torch.empty_like(x3)
    L.payload_call("mgar_bn_act_fwd_grouped", dt, L.pptr(x3, dt), g, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
                   int(relu), L.pptr(y, dt), st)
    return y.view(x.shape)


def bn_train_stats_only(x, bn, per_sample=False):
    """Batch statistics (mean, invstd) of x (G, C, ...) for a train-mode ``bn`` -- running statistics and the batch counter
    updated exactly as a forward pass does -- without applying anything; per_sample: every sample its own statistics, (G * C)
    vectors.  None where the grouped kernel does not apply."""
    x3 = x.contiguous().flatten(2)
    g, c, p = x3.shape
    if not per_sample:
        return _train_stats(x3, bn)
    if g * c > 65535:
        return None
    mean = torch.empty((g * c,), dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    track = bn.track_running_stats and bn.running_mean is not None
    ws = _workspace(x3, 1, g * c, p)
    dt = x3.dtype
    L.payload_call("mgar_bn_train_stats_grouped", dt, L.pptr(x3, dt), g, c, p, float(bn.eps),
                   float(bn.momentum if bn.momentum is not None else 0.1),
                   L.fptr(ws), L.fptr(mean), L.fptr(invstd), L.fptr(bn.running_mean) if track else None,
                   L.fptr(bn.running_var) if track else None,
                   L.dev_ptr(bn.num_batches_tracked, torch.int64) if track and bn.num_batches_tracked is not None else None, L.stream_of(x3))
    return mean, invstd


def bn_apply_with_stats(x, bn, relu, stats, per_sample=False):
    """[relu]((x - mean) * invstd * gamma + beta) with GIVEN statistics (forward only): x may be a different tensor than the one
    the statistics were taken from (Unit3D.forward_then_pool: the pooled tensor)."""
    mean, invstd = stats
    x3 = x.contiguous().flatten(2)
    g, c, p = x3.shape
    gamma, beta = _affine(bn, c, x.device)
    y = torch.empty_like(x3)
    dt = x3.dtype
    if per_sample:
        L.payload_call("mgar_bn_act_fwd_grouped", dt, L.pptr(x3, dt), g, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
                       int(relu), L.pptr(y, dt), L.stream_of(x3))
    else:
        L.payload_call("mgar_bn_act_fwd", dt, L.pptr(x3, dt), g, c, p, L.fptr(mean), L.fptr(invstd), L.fptr(gamma), L.fptr(beta),
                       int(relu), L.pptr(y, dt), L.stream_of(x3))
    return y.view(x.shape)


def bn_act_maxpool(x, bn, relu, in_stats=None):
    """max over the last axis of [relu](bn(x)) for x (B, C, M, ns) -> (B, C, M).  in_stats: statistics partials of x left by
    its producer kernel (stats_partial_buffer)."""
    x4 = x.contiguous()
    if x4.shape[-1] > 255 or (not bn.training and torch.is_grad_enabled() and x.requires_grad):
        return None
    mean, invstd = _stats(x4.flatten(2), bn, in_stats)
    gamma, beta = _affine(bn, x4.shape[1], x.device)
    return _BnActMaxPool.apply(x4, gamma, beta, mean, invstd, relu)[0]
