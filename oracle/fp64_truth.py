"""TEST INFRASTRUCTURE -- float64 ground truth of the op-module COMPOSITIONS; never imported by the product.

The module-level parity tests compare two fp32 evaluations (HIP kernels on the device, C oracle + torch on the host) of
the same composition: train-mode BatchNorm over few columns amplifies fp32 rounding, so those comparisons carry
tolerances looser than north_star's 1e-4 and by themselves cannot say WHICH side is off.  This file
evaluates the same compositions in float64:

* the integer decisions (FPS centres, ball-query rows, 3-NN indices) are the C oracle's, taken from the fp32
  coordinates -- they are part of the reference's definition, bit-exact on both sides, and not what is being measured;
* everything floating point -- relative coordinates, grouping, 3-NN inverse-distance weights, interpolation, the shared
  MLP with train-mode BatchNorm, max-pool -- is float64 torch with autograd;

so that a test can assert  err(HIP vs truth) <= 1.5 * err(oracle backend vs truth).

Compositions follow the reference modules: pointnet2_batch/pointnet2_modules.py:19-55 (SA, MSG) and :122-170 (FP),
pointnet2_stack/pointnet2_modules.py:78-112 (stack SA) and :129-157 (stack FP), with QueryAndGroup as
pointnet2_batch/pointnet2_utils.py:231-264 and pointnet2_stack/pointnet2_utils.py:112-159.
"""
import copy

import numpy as np
import torch

from . import oracle as O


def double_copy(module):
    """Deep copy of an nn.Module with float64 parameters / buffers, on the CPU (its shared MLPs then run as plain torch)."""
    return copy.deepcopy(module).cpu().double()


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().float().numpy())


def _gather_cols(feats, idx):
    """feats (B, C, N), idx (B, M, ns) -> (B, C, M, ns)."""
    b, c, _ = feats.shape
    _, m, ns = idx.shape
    flat = idx.reshape(b, 1, m * ns).expand(b, c, m * ns).long()
    return torch.gather(feats, 2, flat).view(b, c, m, ns)


# ------------------------------------------------------------------------------------------------------ batch layout
def sa_msg_batch(mod64, xyz, features):
    """PointnetSAModuleMSG.forward in float64.  xyz (B, N, 3) fp32-valued, features (B, C, N) float64 (may require grad).
    -> (new_xyz (B, M, 3) float64, (B, sum C_out, M) float64)."""
    xyz32 = _np32(xyz)
    xyz64 = torch.from_numpy(xyz32).double()
    picked, _ = O.fps_batch(xyz32, mod64.npoint)
    pick = torch.from_numpy(picked.astype(np.int64))
    new_xyz64 = torch.gather(xyz64, 1, pick[:, :, None].expand(-1, -1, 3))
    new32 = np.ascontiguousarray(new_xyz64.float().numpy())
    outs = []
    for grouper, mlp in zip(mod64.groupers, mod64.mlps):
        idx = torch.from_numpy(O.ball_query_batch(grouper.radius, grouper.nsample, xyz32, new32).astype(np.int64))
        rel = _gather_cols(xyz64.transpose(1, 2), idx) - new_xyz64.transpose(1, 2)[:, :, :, None]
        if features is not None:
            g = _gather_cols(features, idx)
            grouped = torch.cat([rel, g], 1) if grouper.use_xyz else g
        else:
            grouped = rel
        outs.append(mlp(grouped).max(dim=3).values)
    return new_xyz64, torch.cat(outs, 1)


def _three_nn_weights(unknown32, known32):
    """Indices from the C oracle (fp32 decisions); distances and weights recomputed in float64 from those indices."""
    _, idx = O.three_nn_batch(unknown32, known32)
    idx = torch.from_numpy(idx.astype(np.int64))
    u, k = torch.from_numpy(unknown32).double(), torch.from_numpy(known32).double()
    nb = torch.gather(k[:, None].expand(-1, u.shape[1], -1, -1), 2, idx[:, :, :, None].expand(-1, -1, -1, 3))
    dist = (nb - u[:, :, None, :]).pow(2).sum(-1).sqrt()
    inv = 1.0 / (dist + 1e-8)
    return idx, inv / inv.sum(2, keepdim=True)


def fp_batch(mod64, unknown, known, unknow_feats, known_feats):
    """PointnetFPModule.forward in float64 -> (B, C_out, n)."""
    idx, w = _three_nn_weights(_np32(unknown), _np32(known))
    b, c, _ = known_feats.shape
    n = idx.shape[1]
    g = torch.gather(known_feats, 2, idx.reshape(b, 1, n * 3).expand(b, c, n * 3)).view(b, c, n, 3)
    spread = (g * w[:, None]).sum(-1)
    merged = spread if unknow_feats is None else torch.cat([spread, unknow_feats], 1)
    return mod64.mlp(merged.unsqueeze(-1)).squeeze(-1)


# ------------------------------------------------------------------------------------------------------ stacked layout
def _offsets(cnt):
    c = cnt.detach().cpu().long()
    return torch.cat([c.new_zeros(1), torch.cumsum(c, 0)])


def stack_sa_msg(mod64, xyz, xyz_cnt, new_xyz, new_cnt, features):
    """StackSAModuleMSG.forward in float64.  features (N, C) float64 -> (M, sum C_out) float64."""
    xyz32, new32 = _np32(xyz), _np32(new_xyz)
    xc, nc = xyz_cnt.cpu().numpy().astype(np.int32), new_cnt.cpu().numpy().astype(np.int32)
    xyz64, new64 = torch.from_numpy(xyz32).double(), torch.from_numpy(new32).double()
    start = torch.repeat_interleave(_offsets(xyz_cnt)[:-1], new_cnt.cpu().long())            # first row of each query's sample
    outs = []
    for grouper, mlp in zip(mod64.groupers, mod64.mlps):
        raw = torch.from_numpy(O.ball_query_stack(grouper.radius, grouper.nsample, xyz32, xc, new32, nc).astype(np.int64))
        empty = raw[:, 0] == -1
        rows = (raw.clamp_min(0) + start[:, None])                                              # (M, ns) global rows
        rows = torch.where(empty[:, None], start[:, None].expand_as(rows), rows)
        keep = (~empty).double()[:, None, None]
        rel = (xyz64[rows] - new64[:, None, :]).permute(0, 2, 1) * keep                         # (M, 3, ns)
        if features is not None:
            g = features[rows].permute(0, 2, 1) * keep                                          # (M, C, ns)
            grouped = torch.cat([rel, g], 1) if grouper.use_xyz else g
        else:
            grouped = rel
        x = mlp(grouped.permute(1, 0, 2)[None])                                                 # (1, C', M, ns)
        outs.append(x.max(dim=3).values.squeeze(0).permute(1, 0))
    return torch.cat(outs, 1)


def stack_fp(mod64, unknown, unknown_cnt, known, known_cnt, unknown_feats, known_feats):
    """StackPointnetFPModule.forward in float64 -> (N_unknown, C_out)."""
    u32, k32 = _np32(unknown), _np32(known)
    _, idx = O.three_nn_stack(u32, unknown_cnt.cpu().numpy().astype(np.int32), k32, known_cnt.cpu().numpy().astype(np.int32))
    idx = torch.from_numpy(idx.astype(np.int64))                                                # global rows of `known`
    u, k = torch.from_numpy(u32).double(), torch.from_numpy(k32).double()
    dist = (k[idx] - u[:, None, :]).pow(2).sum(-1).sqrt()
    inv = 1.0 / (dist + 1e-8)
    w = inv / inv.sum(-1, keepdim=True)
    spread = (known_feats[idx] * w[:, :, None]).sum(1)
    merged = spread if unknown_feats is None else torch.cat([spread, unknown_feats], 1)
    x = mod64.mlp(merged.permute(1, 0)[None, :, :, None])
    return x.squeeze(0).squeeze(-1).permute(1, 0)
