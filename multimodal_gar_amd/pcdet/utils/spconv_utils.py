"""The ``spconv`` namespace the reference's sparse trunk is written against (pcdet/utils/spconv_utils.py:3-6 imports the
third-party spconv 2.2.3 -- CUDA-only wheels, not installable on ROCm) on top of this repo's gfx950 sparse-convolution
kernels (multimodal_gar_amd/sparse_ops.py, csrc/sparse_conv.hip).  Only what VoxelBackBone8x touches is provided:

    spconv.SparseConvTensor(features, indices, spatial_shape, batch_size)   .features .indices .spatial_shape .batch_size
                                                                            .replace_feature(f) .dense()
    spconv.SubMConv3d / spconv.SparseConv3d (in, out, kernel_size, stride=1, padding=0, bias=True, indice_key=None)
    spconv.SparseSequential, spconv.SparseModule
    replace_feature(out, new_features), find_all_spconv_keys(model)

Parameter names and shapes follow spconv 2.x (``weight`` (C_out, kz, ky, kx, C_in), optional ``bias`` (C_out)), so a
VoxelBackBone8x state dict of the reference loads unchanged.
"""
import math
from collections import OrderedDict
from typing import Set

import torch
import torch.nn as nn

from ... import sparse_ops


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None, indice_dict=None, benchmark=False):
        self.features = features                      # (N, C)
        self.indices = indices.int().contiguous()     # (N, 4) [b, z, y, x]
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict     # indice_key -> rulebook, shared along the lineage

    def replace_feature(self, feature):
        out = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, indice_dict=self.indice_dict)
        return out

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def dense(self, channels_first=True):
        c = self.features.shape[1]
        z, y, x = self.spatial_shape
        out = self.features.new_zeros((self.batch_size, z, y, x, c))
        idx = self.indices.long()
        out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out


class SparseModule(nn.Module):
    """Marker base class: modules that take and return a SparseConvTensor."""


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 subm=False, indice_key=None, **kwargs):
        super().__init__()
        assert ndim == 3 and groups == 1 and sparse_ops._triple(dilation) == (1, 1, 1)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = sparse_ops._triple(kernel_size), sparse_ops._triple(stride), sparse_ops._triple(padding)
        self.subm, self.indice_key = subm, indice_key
        if subm:   # spconv ignores stride / padding of a submanifold convolution: output sites = input sites
            self.stride, self.padding = (1, 1, 1), tuple(k // 2 for k in self.kernel_size)
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: SparseConvTensor):
        feats, idx, shape = sparse_ops.sparse_conv3d(x.features, x.indices, x.spatial_shape, x.batch_size, self.weight, self.kernel_size,
                                                     self.stride, self.padding, self.subm, x.indice_dict, self.indice_key)
        if self.bias is not None:
            feats = feats + self.bias
        return SparseConvTensor(feats, idx, shape, x.batch_size, indice_dict=x.indice_dict)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None,
                 **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True, indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None,
                 **kwargs):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, False, indice_key)


class SparseSequential(SparseModule):
    """nn.Sequential for mixed sparse / dense modules: dense ones (BatchNorm1d, ReLU) act on ``.features``."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for i, module in enumerate(args):
                self.add_module(str(i), module)
        for name, module in kwargs.items():
            self.add_module(name, module)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def __len__(self):
        return len(self._modules)

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if isinstance(module, SparseModule):
                x = module(x)
            elif isinstance(x, SparseConvTensor):
                feats = x.features
                if isinstance(module, nn.BatchNorm1d) and feats.is_cuda:
                    # BatchNorm1d [+ ReLU] on the row-major (N_active, C) features in the library's row-major kernels (one
                    # statistics pass, one apply pass; backward likewise) instead of torch's native batch norm + threshold
                    from ... import bn_ops
                    relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                    y = bn_ops.bn_act_rows(feats, module, relu)
                    if y is not None:
                        x = x.replace_feature(y)
                        i += 2 if relu else 1
                        continue
                x = x.replace_feature(module(feats))
            else:
                x = module(x)
            i += 1
        return x


class _Namespace:
    SparseConvTensor = SparseConvTensor
    SparseModule = SparseModule
    SparseSequential = SparseSequential
    SubMConv3d = SubMConv3d
    SparseConv3d = SparseConv3d

    class conv:
        SparseConvolution = SparseConvolution


spconv = _Namespace


def find_all_spconv_keys(model: nn.Module, prefix="") -> Set[str]:
    """Names of the sparse-convolution weights of `model` (reference spconv_utils.py:11-27)."""
    found: Set[str] = set()
    for name, child in model.named_children():
        new_prefix = "%s.%s" % (prefix, name) if prefix else name
        if isinstance(child, SparseConvolution):
            found.add(new_prefix + ".weight")
        found.update(find_all_spconv_keys(child, prefix=new_prefix))
    return found


def replace_feature(out, new_features):
    return out.replace_feature(new_features)
