"""RGB backbone blocks: Inception-I3D and the non-local block.

Mirror of the public surface of the reference's model/backbone.py (class names, constructor
and forward signatures, sub-module / parameter names and shapes, so that the Kinetics
``rgb_imagenet.pt`` checkpoint the reference loads at model/gat_model.py:990-991 and the
reference's own checkpoints load here unchanged):
  MaxPool3dSamePadding (backbone.py:99-131), Unit3D (:134-206), InceptionModule (:210-235),
  InceptionI3d (:238-425), NLBlockND (:558-687).
The dense convolutions stay on PyTorch-ROCm / MIOpen: they are
GEMM-shaped library work, not irregular-index kernels.

Difference in mechanism, not in result: TensorFlow-"same" padding is symmetric whenever the
total pad is even (every stride-1 odd-kernel layer); in that case the pad is handed to the
convolution itself instead of materialising a padded copy of the activation with F.pad.
"""
from typing import Sequence, Tuple

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nn_utils import PointwiseSequential, conv1x1


def _same_pad_1d(size: int, kernel: int, stride: int) -> Tuple[int, int]:
    """(front, back) padding of TF 'same' mode (backbone.py:101-105, :123-128)."""
    total = max(kernel - stride, 0) if size % stride == 0 else max(kernel - (size % stride), 0)
    front = total // 2
    return front, total - front


def _same_pads(shape_thw: Sequence[int], kernel: Sequence[int], stride: Sequence[int]):
    return [_same_pad_1d(s, k, st) for s, k, st in zip(shape_thw, kernel, stride)]


def _as_fpad(pads):
    # F.pad order: last dim first -> (w_f, w_b, h_f, h_b, t_f, t_b)
    (tf, tb), (hf, hb), (wf, wb) = pads
    return (wf, wb, hf, hb, tf, tb)


class MaxPool3dSamePadding(nn.MaxPool3d):
    def compute_pad(self, dim, s):
        return sum(_same_pad_1d(s, self.kernel_size[dim], self.stride[dim]))

    def forward(self, x):
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and not (torch.is_grad_enabled() and x.requires_grad):
            # forward-only fused kernel (csrc/maxpool3d.hip): no padded copy, no index tensor; fp32 or bf16 payload
            from .. import _lib as L
            n, c, t, h, w = x.shape
            (kt, kh, kw), (st, sh, sw) = self.kernel_size, self.stride
            if x.dim() == 5 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last_3d) and c % 4 == 0:
                # channels-last activations stay channels-last (csrc/channels_last.hpp)
                y = torch.empty((n, c, -(-t // st), -(-h // sh), -(-w // sw)), dtype=x.dtype, device=x.device,
                                memory_format=torch.channels_last_3d)
                L.payload_call("mgar_maxpool3d_same_fwd_cl", x.dtype, x.data_ptr(), n, t, h, w, c, kt, kh, kw, st, sh, sw, y.data_ptr(),
                               L.stream_of(x))
                return y
            x = x.contiguous()
            y = torch.empty((n, c, -(-t // st), -(-h // sh), -(-w // sw)), dtype=x.dtype, device=x.device)
            L.payload_call("mgar_maxpool3d_same_fwd", x.dtype, L.pptr(x, x.dtype), n * c, t, h, w, kt, kh, kw, st, sh, sw,
                           L.pptr(y, x.dtype), L.stream_of(x))
            return y
        pads = _same_pads(x.shape[2:], self.kernel_size, self.stride)
        return super().forward(F.pad(x, _as_fpad(pads)))  # zero pad, as the reference

    def forward_valid(self, x):
        """The same windows, maximum over their VALID elements only (csrc/maxpool3d.hip, mgar_maxpool3d_valid_fwd): what
        Unit3D.forward_then_pool pools the PRE-BatchNorm tensor with.  Device, NCDHW, forward only."""
        from .. import _lib as L
        x = x.contiguous()
        n, c, t, h, w = x.shape
        (kt, kh, kw), (st, sh, sw) = self.kernel_size, self.stride
        y = torch.empty((n, c, -(-t // st), -(-h // sh), -(-w // sw)), dtype=x.dtype, device=x.device)
        L.payload_call("mgar_maxpool3d_valid_fwd", x.dtype, L.pptr(x, x.dtype), n * c, t, h, w, kt, kh, kw, st, sh, sw,
                       L.pptr(y, x.dtype), L.stream_of(x))
        return y


class Unit3D(nn.Module):
    """Conv3d (no bias by default) + BatchNorm3d(eps=1e-3, momentum=0.01) + ReLU with dynamic
    'same' padding."""

    per_sample_stats = False   # see InceptionI3d.set_per_sample_stats
    emit_channels_last = False  # see InceptionI3d.set_channels_last: this unit's BatchNorm writes NDHWC (the stem)

    def __init__(self, in_channels, output_channels, kernel_shape=(1, 1, 1), stride=(1, 1, 1), padding=0,
                 activation_fn=F.relu, use_batch_norm=True, use_bias=False, name='unit_3d'):
        super().__init__()
        self._output_channels = output_channels
        self._kernel_shape = tuple(kernel_shape)
        self._stride = tuple(stride)
        self._use_batch_norm = use_batch_norm
        self._activation_fn = activation_fn
        self._use_bias = use_bias
        self.name = name
        self.padding = padding
        self.conv3d = nn.Conv3d(in_channels, output_channels, kernel_size=self._kernel_shape, stride=self._stride,
                                padding=0, bias=use_bias)
        if use_batch_norm:
            self.bn = nn.BatchNorm3d(output_channels, eps=0.001, momentum=0.01)

    def compute_pad(self, dim, s):
        return sum(_same_pad_1d(s, self._kernel_shape[dim], self._stride[dim]))

    stem_kernel = True     # 3 -> 64, 7x7x7, stride 2 on the device: csrc/stem_conv.hip instead of pad + library convolution

    def _stem_conv(self, x):
        """The I3D stem (Conv3d_1a_7x7) on csrc/stem_conv.hip: "same" padding inside the kernel, NCDHW in and out; or None."""
        c = self.conv3d
        if not (self.stem_kernel and x.is_cuda and x.dim() == 5 and x.dtype in (torch.float32, torch.bfloat16)
                and c.in_channels == 3 and c.out_channels == 64 and self._kernel_shape == (7, 7, 7) and self._stride == (2, 2, 2)
                and c.bias is None and not (torch.is_grad_enabled() and (x.requires_grad or c.weight.requires_grad))):
            return None
        from .. import _lib as L
        if torch.is_autocast_enabled() and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype('cuda'))
        x = x.contiguous()
        n, _, t, h, w = x.shape
        y = torch.empty((n, 64, (t + 1) // 2, (h + 1) // 2, (w + 1) // 2), dtype=x.dtype, device=x.device)
        wp = torch.empty((L.raw("mgar_stem_conv3d_workspace_floats"),), dtype=torch.float32, device=x.device)
        wf = c.weight.detach().float().contiguous()
        L.payload_call("mgar_stem_conv3d_fwd", x.dtype, L.pptr(x, x.dtype), n, t, h, w, L.fptr(wf), L.fptr(wp), L.pptr(y, x.dtype),
                       L.stream_of(x))
        return y

    gemm_1x1 = os.environ.get("MGAR_I3D_GEMM_1X1", "1") != "0"        # 1x1x1 units on the device: library GEMMs (one per sample) instead of the library convolution
    wino_kernel = os.environ.get("MGAR_I3D_OWN_CONV", "1") != "0"     # 3x3x3, stride 1 on the device: csrc/conv3d_wino.hip (Winograd F(2,3) along W on the fp32 MFMA)

    def _k3_conv(self, x):
        """The 3x3x3 / stride-1 units (Conv3d_2c_3x3, every Mixed block's Conv3d_0b_3x3) on csrc/conv3d_wino.hip: "same"
        padding inside the kernel, NCDHW in and out, fp32, forward only; or None (the library convolution then)."""
        c = self.conv3d
        if not (self.wino_kernel and x.is_cuda and x.dim() == 5 and x.dtype == torch.float32 and not torch.is_autocast_enabled()
                and self._kernel_shape == (3, 3, 3) and self._stride == (1, 1, 1) and c.bias is None and c.groups == 1
                and c.in_channels % 2 == 0 and x.shape[4] % 2 == 0 and x.is_contiguous() and c.weight.dtype == torch.float32
                and not (torch.is_grad_enabled() and (x.requires_grad or c.weight.requires_grad))):
            return None
        from .. import _lib as L
        n, cin, d, h, w = x.shape
        y = torch.empty((n, c.out_channels, d, h, w), dtype=torch.float32, device=x.device)
        wp = torch.empty((L.raw("mgar_conv3d_k3_workspace_floats", cin, c.out_channels),), dtype=torch.float32, device=x.device)
        wf = c.weight.detach().contiguous()
        L.call("mgar_conv3d_k3_fwd", L.fptr(x), n, cin, d, h, w, L.fptr(wf), c.out_channels, L.fptr(wp), L.fptr(y), L.stream_of(x))
        return y

    def _conv(self, x):
        pads = _same_pads(x.shape[2:], self._kernel_shape, self._stride)
        stem = self._stem_conv(x)
        if stem is not None:
            return stem
        k3 = self._k3_conv(x)
        if k3 is not None:
            return k3
        if (self.gemm_1x1 and x.is_cuda and x.dim() == 5 and self._kernel_shape == (1, 1, 1) and self._stride == (1, 1, 1)
                and x.is_contiguous() and self.conv3d.groups == 1 and not torch.is_autocast_enabled()
                and x.dtype == self.conv3d.weight.dtype
                and not (torch.is_grad_enabled() and (x.requires_grad or self.conv3d.weight.requires_grad))):
            # a 1x1x1 convolution is the GEMM W (C_out, C_in) x (C_in, T*H*W) per sample, on the NCDHW tensor as it lies (MIOpen
            # wraps its NDHWC kernels in two layout transposes and needs a find pass per shape).  One plain GEMM per sample, not one
            # strided-batched call: with 5 clips per pass the batched route hung the two-stream graph replay in 2 of 4 runs (4 of 4
            # with a materialised batch of weights), the per-sample route in 0 of 4 at the same speed -- two library GEMMs of the
            # batched kind running on two streams at once is the only combination that ever hung (known issues).
            c = self.conv3d
            w2 = c.weight.view(c.out_channels, c.in_channels)
            x3 = x.flatten(2)
            if os.environ.get("MGAR_1X1_MODE", "mm_loop") == "bmm":          # diagnostics: the batched call
                y = torch.bmm(w2.unsqueeze(0).expand(x3.shape[0], -1, -1), x3)
            else:
                y = torch.empty((x3.shape[0], c.out_channels, x3.shape[2]), dtype=x.dtype, device=x.device)
                for n in range(x3.shape[0]):
                    torch.mm(w2, x3[n], out=y[n])
            if c.bias is not None:
                y = y + c.bias.view(1, -1, 1)
            return y.view(x.shape[0], c.out_channels, *x.shape[2:])
        if all(f == b for f, b in pads):
            return F.conv3d(x, self.conv3d.weight, self.conv3d.bias, self._stride, tuple(f for f, _ in pads))
        return self.conv3d(F.pad(x, _as_fpad(pads)))

    pool_first = True      # forward_then_pool: normalise the pooled tensor instead of the full one (frozen device path)

    def _gamma_positive(self):
        """gamma > 0 in every channel (what makes relu(bn(.)) monotone non-decreasing); one host read per weight version --
        the frozen I3D's weights never change, and the first call happens in an eager warm-up step, never inside a capture."""
        w = self.bn.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, "_gamma_pos_key", None) != key:
            self._gamma_pos_key, self._gamma_pos = key, bool((w.detach() > 0).all().item())
        return self._gamma_pos

    def forward_then_pool(self, x, pool):
        """pool(self(x)) for a MaxPool3dSamePadding ``pool`` that directly follows this unit (reference model/backbone.py:305-313:
        Conv3d_1a_7x7 -> MaxPool3d_2a_3x3, Conv3d_2c_3x3 -> MaxPool3d_3a_3x3), or None where that does not apply.
        relu(bn(.)) is monotone non-decreasing per channel when gamma > 0 (in fp32 as well) and >= 0, so
        maxpool_same(relu(bn(z))) == relu(bn(maxpool_valid(z))) bit for bit: the batch statistics come from the FULL
        pre-BatchNorm tensor z as before, but the normalisation + ReLU pass runs over the pooled tensor (a quarter of z for the
        stem) -- one read + one write of z less (7.6 GB of the stem's output per c3 step)."""
        if not (self.pool_first and self._use_batch_norm and self._activation_fn is F.relu and self.bn.training and self.bn.affine
                and x.is_cuda and x.dim() == 5 and not self.emit_channels_last
                and not (torch.is_grad_enabled() and (x.requires_grad or self.bn.weight.requires_grad))):
            return None
        if torch.cuda.is_current_stream_capturing() and getattr(self, "_gamma_pos_key", None) is None:
            return None
        if not self._gamma_positive():
            return None
        from .. import bn_ops
        z = self._conv(x)
        if z.dtype not in (torch.float32, torch.bfloat16) or not z.is_contiguous():
            return pool(self._bn_relu(z))
        per = z.shape[2] * z.shape[3] * z.shape[4] * (1 if (self.per_sample_stats and z.shape[0] > 1) else z.shape[0])
        if per <= bn_ops.SMALL_CHANNEL_MAX:      # small tensors: the one-launch statistics + apply kernel of the plain path (nothing to gain here,
            return pool(self._bn_relu(z))        # and its statistics arithmetic differs in the last bits)
        stats = bn_ops.bn_train_stats_only(z, self.bn, per_sample=self.per_sample_stats and z.shape[0] > 1)
        if stats is None:
            return pool(self._bn_relu(z))
        return bn_ops.bn_apply_with_stats(pool.forward_valid(z), self.bn, True, stats, per_sample=self.per_sample_stats and z.shape[0] > 1)

    def forward(self, x, out=None):
        """``out`` (optional): a channel slice y[:, c0:c1] of a wider tensor the result should land in (the caller's
        concatenation); honoured where the fused BatchNorm + ReLU kernel writes the result, ignored otherwise --
        the caller checks ``result is out``."""
        return self._bn_relu(self._conv(x), out)

    def _bn_relu(self, x, out=None):
        relu_fused = False
        if self._use_batch_norm:
            y = None
            if x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad and not self.bn.training):
                # BatchNorm3d + ReLU in one streaming pass of csrc/bn_act.hip (the reference runs them
                # as two kernels; I3D is frozen, so this is forward-only in MGAR-net)
                from .. import bn_ops
                relu_fused = self._activation_fn is F.relu
                if self.per_sample_stats and x.shape[0] > 1:
                    # several clips in one pass, each normalised with its own statistics = one pass per clip
                    y = bn_ops.bn_act_per_sample(x, self.bn, relu_fused, out=out, to_channels_last=self.emit_channels_last)
                    assert y is not None, "per_sample_stats is a forward-only (frozen backbone) device path"
                else:
                    y = bn_ops.bn_act(x, self.bn, relu_fused, out=out, to_channels_last=self.emit_channels_last)
            x = self.bn(x) if y is None else y
            relu_fused = relu_fused and y is not None
        if self._activation_fn is not None and not relu_fused:
            x = self._activation_fn(x)
        return x


class InceptionModule(nn.Module):
    def __init__(self, in_channels, out_channels, name):
        super().__init__()
        oc = out_channels
        self.b0 = Unit3D(in_channels, oc[0], [1, 1, 1], name=name + '/Branch_0/Conv3d_0a_1x1')
        self.b1a = Unit3D(in_channels, oc[1], [1, 1, 1], name=name + '/Branch_1/Conv3d_0a_1x1')
        self.b1b = Unit3D(oc[1], oc[2], [3, 3, 3], name=name + '/Branch_1/Conv3d_0b_3x3')
        self.b2a = Unit3D(in_channels, oc[3], [1, 1, 1], name=name + '/Branch_2/Conv3d_0a_1x1')
        self.b2b = Unit3D(oc[3], oc[4], [3, 3, 3], name=name + '/Branch_2/Conv3d_0b_3x3')
        self.b3a = MaxPool3dSamePadding(kernel_size=[3, 3, 3], stride=(1, 1, 1), padding=0)
        self.b3b = Unit3D(in_channels, oc[5], [1, 1, 1], name=name + '/Branch_3/Conv3d_0b_1x1')
        self.name = name

    def forward(self, x):
        if x.is_cuda and not (torch.is_grad_enabled() and x.requires_grad):
            # frozen backbone on the device: every branch's last BatchNorm + ReLU writes its channels of the concatenated
            # output directly (csrc/bn_act.hip, mgar_bn_act_fwd_into): the torch.cat pass (a read and a write of the whole
            # module output, 3.6 ms per step at config c3) disappears
            ends = (self.b0, self.b1b, self.b2b, self.b3b)
            widths = [u.conv3d.out_channels for u in ends]
            dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else x.dtype
            cl = not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last_3d)
            y = torch.empty((x.shape[0], sum(widths)) + tuple(x.shape[2:]), dtype=dt, device=x.device,   # 1x1x1 / "same" 3x3x3, stride 1
                            memory_format=torch.channels_last_3d if cl else torch.contiguous_format)
            c0 = 0
            for k, (unit, w) in enumerate(zip(ends, widths)):
                h = x if k == 0 else (self.b1a(x) if k == 1 else (self.b2a(x) if k == 2 else self.b3a(x)))
                dst = y[:, c0:c0 + w]
                r = unit(h, out=dst)
                if r is not dst:                                 # a path that did not take `out` (shape / dtype / layout): one copy
                    dst.copy_(r)
                c0 += w
            return y
        return torch.cat([self.b0(x), self.b1b(self.b1a(x)), self.b2b(self.b2a(x)), self.b3b(self.b3a(x))], dim=1)


# (endpoint, kind, spec).  Channel plan of Inception-v1 I3D (backbone.py:305-375).
_I3D_PLAN = (
    ('Conv3d_1a_7x7', 'conv', dict(cin=None, cout=64, k=[7, 7, 7], s=(2, 2, 2), pad=(3, 3, 3))),
    ('MaxPool3d_2a_3x3', 'pool', dict(k=[1, 3, 3], s=(1, 2, 2))),
    ('Conv3d_2b_1x1', 'conv', dict(cin=64, cout=64, k=[1, 1, 1], s=(1, 1, 1), pad=0)),
    ('Conv3d_2c_3x3', 'conv', dict(cin=64, cout=192, k=[3, 3, 3], s=(1, 1, 1), pad=1)),
    ('MaxPool3d_3a_3x3', 'pool', dict(k=[1, 3, 3], s=(1, 2, 2))),
    ('Mixed_3b', 'mixed', dict(cin=192, oc=[64, 96, 128, 16, 32, 32])),
    ('Mixed_3c', 'mixed', dict(cin=256, oc=[128, 128, 192, 32, 96, 64])),
    ('MaxPool3d_4a_3x3', 'pool', dict(k=[3, 3, 3], s=(2, 2, 2))),
    ('Mixed_4b', 'mixed', dict(cin=480, oc=[192, 96, 208, 16, 48, 64])),
    ('Mixed_4c', 'mixed', dict(cin=512, oc=[160, 112, 224, 24, 64, 64])),
    ('Mixed_4d', 'mixed', dict(cin=512, oc=[128, 128, 256, 24, 64, 64])),
    ('Mixed_4e', 'mixed', dict(cin=512, oc=[112, 144, 288, 32, 64, 64])),
    ('Mixed_4f', 'mixed', dict(cin=528, oc=[256, 160, 320, 32, 128, 128])),
    ('MaxPool3d_5a_2x2', 'pool', dict(k=[2, 2, 2], s=(2, 2, 2))),
    ('Mixed_5b', 'mixed', dict(cin=832, oc=[256, 160, 320, 32, 128, 128])),
    ('Mixed_5c', 'mixed', dict(cin=832, oc=[384, 192, 384, 48, 128, 128])),
)


class InceptionI3d(nn.Module):
    """Inception-v1 I3D (Carreira & Zisserman 2017).  MGAR-net builds it up to 'Mixed_4f'
    (832 channels, stride (4, 16, 16)); ``extract_features`` runs the built endpoints."""

    VALID_ENDPOINTS = tuple(p[0] for p in _I3D_PLAN) + ('Logits', 'Predictions')

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint='Logits', name='inception_i3d',
                 in_channels=3, dropout_keep_prob=0.5):
        if final_endpoint not in self.VALID_ENDPOINTS:
            raise ValueError('Unknown final endpoint %s' % final_endpoint)
        super().__init__()
        self._num_classes = num_classes
        self._spatial_squeeze = spatial_squeeze
        self._final_endpoint = final_endpoint
        self.logits = None
        self.end_points = {}
        for end_point, kind, sp in _I3D_PLAN:
            if kind == 'conv':
                layer = Unit3D(in_channels if sp['cin'] is None else sp['cin'], sp['cout'], sp['k'], sp['s'], sp['pad'],
                               name=name + end_point)
            elif kind == 'pool':
                layer = MaxPool3dSamePadding(kernel_size=sp['k'], stride=sp['s'], padding=0)
            else:
                layer = InceptionModule(sp['cin'], sp['oc'], name + end_point)
            self.end_points[end_point] = layer
            if self._final_endpoint == end_point:
                return  # like the reference: the caller must call build() (gat_model.py:988)
        self.avg_pool = nn.AvgPool3d(kernel_size=[2, 7, 7], stride=(1, 1, 1))
        self.dropout = nn.Dropout(dropout_keep_prob)
        self.replace_logits(num_classes)
        self.build()

    def replace_logits(self, num_classes):
        self._num_classes = num_classes
        self.logits = Unit3D(1024, num_classes, [1, 1, 1], padding=0, activation_fn=None, use_batch_norm=False,
                             use_bias=True, name='logits')

    def build(self):
        for k, layer in self.end_points.items():
            self.add_module(k, layer)

    def set_channels_last(self, on=True):
        """Keep the activations NDHWC between the stem and the output (frozen, forward-only device path): MIOpen's
        composable-kernel convolutions work in that layout and otherwise wrap every call in two transposes (46 launches,
        5 ms per step at config c3).  The stem's BatchNorm changes the layout; the convolution weights are stored
        channels-last once (same values, same state dict)."""
        first = True
        for end_point, layer in self.end_points.items():
            for m in layer.modules():
                if isinstance(m, Unit3D):
                    if first:
                        m.emit_channels_last = bool(on)
                        first = False
                    elif m.conv3d.weight.dim() == 5:
                        fmt = torch.channels_last_3d if on else torch.contiguous_format
                        m.conv3d.weight.data = m.conv3d.weight.data.contiguous(memory_format=fmt)
        self.channels_last = bool(on)

    def set_per_sample_stats(self, on=True):
        """Train-mode BatchNorm statistics per SAMPLE instead of per batch (device, forward-only): a batch of clips
        then gives exactly what the reference's one-clip-at-a-time passes give (gat_model.py:1048), in one pass."""
        for m in self.modules():
            if isinstance(m, Unit3D):
                m.per_sample_stats = bool(on)
        return self

    def extract_features(self, x):
        names = [e for e in self.VALID_ENDPOINTS if e in self.end_points]
        i = 0
        while i < len(names):
            layer = self._modules[names[i]]
            nxt = self._modules[names[i + 1]] if i + 1 < len(names) else None
            # (a forward hook on either module wants that module's own output: take the plain path then)
            if isinstance(layer, Unit3D) and isinstance(nxt, MaxPool3dSamePadding) and not (layer._forward_hooks or nxt._forward_hooks):
                y = layer.forward_then_pool(x, nxt)          # BatchNorm + ReLU after the pooling (same values): device, frozen
                if y is not None:
                    x = y
                    i += 2
                    continue
            x = layer(x)
            i += 1
        return x

    def forward(self, x):
        x = self.logits(self.dropout(self.avg_pool(self.extract_features(x))))
        return x.squeeze(3).squeeze(3) if self._spatial_squeeze else x


class NLBlockND(nn.Module):
    """Non-local block (Wang et al. 2018) without sub-sampling; modes gaussian / embedded / dot /
    concatenate.  MGAR-net uses mode='dot': f = theta^T phi / N, no softmax (backbone.py:673-675).
    With bn_layer the BN after W_z starts at gamma = beta = 0, i.e. the block is the identity at
    initialisation (:612-614)."""

    def __init__(self, in_channels, inter_channels=None, mode='embedded', dimension=3, bn_layer=True):
        super().__init__()
        assert dimension in [1, 2, 3]
        if mode not in ['gaussian', 'embedded', 'dot', 'concatenate']:
            raise ValueError('`mode` must be one of `gaussian`, `embedded`, `dot` or `concatenate`')
        self.mode, self.dimension, self.in_channels = mode, dimension, in_channels
        self.inter_channels = inter_channels if inter_channels is not None else max(in_channels // 2, 1)
        conv_nd = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dimension]
        bn = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}[dimension]
        self.g = conv_nd(self.in_channels, self.inter_channels, kernel_size=1)
        if bn_layer:
            self.W_z = PointwiseSequential(conv_nd(self.inter_channels, self.in_channels, kernel_size=1), bn(self.in_channels))
            nn.init.constant_(self.W_z[1].weight, 0)
            nn.init.constant_(self.W_z[1].bias, 0)
        else:
            self.W_z = conv_nd(self.inter_channels, self.in_channels, kernel_size=1)
            nn.init.constant_(self.W_z.weight, 0)
            nn.init.constant_(self.W_z.bias, 0)
        if mode in ('embedded', 'dot', 'concatenate'):
            self.theta = conv_nd(self.in_channels, self.inter_channels, kernel_size=1)
            self.phi = conv_nd(self.in_channels, self.inter_channels, kernel_size=1)
        if mode == 'concatenate':
            self.W_f = nn.Sequential(nn.Conv2d(self.inter_channels * 2, 1, kernel_size=1), nn.ReLU())

    def forward(self, x):
        """x: (N, C, T, H, W) / (N, C, H, W) / (N, C, T) for dimension 3 / 2 / 1."""
        n = x.size(0)
        g_x = conv1x1(self.g, x).view(n, self.inter_channels, -1).permute(0, 2, 1)  # (N, P, Ci)
        if self.mode == 'gaussian':
            theta_x = x.view(n, self.in_channels, -1).permute(0, 2, 1)
            f = torch.matmul(theta_x, x.view(n, self.in_channels, -1))
        elif self.mode == 'dot' and g_x.shape[1] > self.inter_channels:
            # f = theta^T phi / P has no softmax, so y = (theta^T phi / P) g = theta^T ((phi g) / P): the (P, P) matrix -- 716 MB
            # per pass for the 6^3 RoI grids of config c3 -- is never formed; Ci x Ci products instead (same algebra, the
            # sums in another order)
            theta_x = conv1x1(self.theta, x).view(n, self.inter_channels, -1).permute(0, 2, 1)
            phi_g = torch.matmul(conv1x1(self.phi, x).view(n, self.inter_channels, -1), g_x) / g_x.shape[1]   # (N, Ci, Ci)
            y = torch.matmul(theta_x, phi_g).permute(0, 2, 1).contiguous().view(n, self.inter_channels, *x.size()[2:])
            return (self.W_z(y) if isinstance(self.W_z, nn.Sequential) else conv1x1(self.W_z, y)) + x
        elif self.mode in ('embedded', 'dot'):
            theta_x = conv1x1(self.theta, x).view(n, self.inter_channels, -1).permute(0, 2, 1)
            f = torch.matmul(theta_x, conv1x1(self.phi, x).view(n, self.inter_channels, -1))  # (N, P, P)
        else:
            theta_x = self.theta(x).view(n, self.inter_channels, -1, 1)
            phi_x = self.phi(x).view(n, self.inter_channels, 1, -1)
            h, w = theta_x.size(2), phi_x.size(3)
            f = self.W_f(torch.cat([theta_x.repeat(1, 1, 1, w), phi_x.repeat(1, 1, h, 1)], dim=1))
            f = f.view(f.size(0), f.size(2), f.size(3))
        if self.mode in ('gaussian', 'embedded'):
            f_div_c = F.softmax(f, dim=-1)
        else:
            f_div_c = f / f.size(-1)
        y = torch.matmul(f_div_c, g_x).permute(0, 2, 1).contiguous()
        y = y.view(n, self.inter_channels, *x.size()[2:])
        return (self.W_z(y) if isinstance(self.W_z, nn.Sequential) else conv1x1(self.W_z, y)) + x


def _needs_torchvision(name):
    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            raise ImportError('%s wraps torchvision.models (reference backbone.py:7-96); torchvision is not '
                              'available and these ImageNet backbones are not on the MGAR-net hot path' % name)
    _Missing.__name__ = name
    return _Missing


MyInception_v3 = _needs_torchvision('MyInception_v3')
MyVGG16 = _needs_torchvision('MyVGG16')
MyVGG19 = _needs_torchvision('MyVGG19')
