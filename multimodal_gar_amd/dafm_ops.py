"""Fused DAFM attention core (HIP): E = softmax(-De/sigma), Att = softmax(QK^T * E * scale),
out = Att V -- model/gat_model.py:487-491, :503-505 -- batched over scenes."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L


class _DafmAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v, de_flat, scene_off, de_off, sigma, scale):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        rows, d = q.shape
        n_scenes = scene_off.numel() - 1
        att = torch.empty_like(de_flat)
        out = torch.empty_like(q)
        L.call("mgar_dafm_attn_fwd", n_scenes, rows, d, L.iptr(scene_off), L.iptr(de_off), L.fptr(q), L.fptr(k),
               L.fptr(v), L.fptr(de_flat), float(sigma), float(scale), L.fptr(att), L.fptr(out), L.stream_of(q))
        ctx.save_for_backward(q, k, v, de_flat, scene_off, de_off, att)
        ctx.cfg = (float(sigma), float(scale))
        return out, att

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, grad_att_unused):
        q, k, v, de_flat, scene_off, de_off, att = ctx.saved_tensors
        sigma, scale = ctx.cfg
        rows, d = q.shape
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        gmat = torch.empty_like(att)
        grad_out = grad_out.contiguous()
        L.call("mgar_dafm_attn_bwd", scene_off.numel() - 1, rows, d, L.iptr(scene_off), L.iptr(de_off), L.fptr(q),
               L.fptr(k), L.fptr(v), L.fptr(de_flat), sigma, scale, L.fptr(att), L.fptr(grad_out),
               L.fptr(gmat), L.fptr(gq), L.fptr(gk), L.fptr(gv), L.stream_of(q))
        return gq, gk, gv, None, None, None, None, None


_OFFSET_CACHE = {}


def scene_offsets(counts, device):
    """counts: python list of actors per scene -> (scene_off (S+1,), de_off (S,)) int32 on device.
    Cached per (counts, device): no host->device copy in the steady state (and none under graph capture)."""
    key = (tuple(int(c) for c in counts), str(device))
    hit = _OFFSET_CACHE.get(key)
    if hit is not None:
        return hit
    so, do, r, m = [0], [], 0, 0
    for n in counts:
        do.append(m)
        r += n
        m += n * n
        so.append(r)
    out = (torch.tensor(so, dtype=torch.int32, device=device), torch.tensor(do, dtype=torch.int32, device=device))
    if len(_OFFSET_CACHE) < 64:
        _OFFSET_CACHE[key] = out
    return out


def dafm_attention(q, k, v, de_flat, scene_off, de_off, sigma, scale):
    """q, k, v: (rows, D) stacked over scenes; de_flat: concatenation of each scene's (n, n)
    distance matrix, row-major.  Returns (out (rows, D), att (same layout as de_flat))."""
    # The per-scene attention tensors are tiny (A x 512 per scene): they stay fp32 on every configuration; under the bf16
    # configurations (autocast) the projections that produce q / k / v run as bf16 GEMMs and are widened here.
    return _DafmAttention.apply(q.float(), k.float(), v.float(), de_flat.float(), scene_off, de_off, sigma, scale)
