"""GATv2Conv on the HIP edge-softmax/aggregate kernel.

Drop-in for ``torch_geometric.nn.GATv2Conv`` as the reference constructs and calls it
(model/gat_model.py:1019  GATv2Conv(512, 512, 8, dropout=0.5, concat=False);  :1094
``GAT_module(x, edge_index)``).  torch_geometric is an unpinned third-party dependency that
is absent from the reference tree and from this image; parameter names and shapes follow
PyG 2.x (lin_l, lin_r: Linear(in, heads*out, bias=True); att: (1, heads, out); bias: (out)
when concat=False) so a PyG state dict loads, and the arithmetic follows the published layer
(SURVEY.md section 8c)."""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L


def glorot_(t: torch.Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class _GatAggregate(Function):
    @staticmethod
    def forward(ctx, xl, xr, att, rowptr, col, edge_scale, heads, slope, by_source=None):
        xl, xr, att = xl.contiguous(), xr.contiguous(), att.contiguous()
        n = xl.shape[0]
        c = xl.shape[1] // heads
        alpha = torch.empty((col.numel(), heads), dtype=torch.float32, device=xl.device)
        out = torch.empty_like(xl)
        L.call("mgar_gatv2_fwd", n, heads, c, L.iptr(rowptr), L.iptr(col), L.fptr(xl), L.fptr(xr), L.fptr(att),
               float(slope), L.fptr(edge_scale) if edge_scale is not None else None, L.fptr(alpha), L.fptr(out),
               L.stream_of(xl))
        if by_source is None:
            by_source = csr_by_source(rowptr, col)
        ctx.save_for_backward(xl, xr, att, rowptr, col, alpha, edge_scale if edge_scale is not None else torch.empty(0), *by_source)
        ctx.cfg = (heads, float(slope), edge_scale is not None)
        return out, alpha

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, grad_alpha_unused):
        xl, xr, att, rowptr, col, alpha, edge_scale, src_rowptr, src_edge, src_dst = ctx.saved_tensors
        heads, slope, has_scale = ctx.cfg
        n = xl.shape[0]
        c = xl.shape[1] // heads
        gxl, gxr, gatt = torch.empty_like(xl), torch.empty_like(xr), torch.empty_like(att)
        grad_out = grad_out.contiguous()
        ws = torch.empty((L.raw("mgar_gatv2_bwd_workspace_floats", n, heads, c, col.numel()),), dtype=torch.float32, device=xl.device)
        L.call("mgar_gatv2_bwd", n, heads, c, L.iptr(rowptr), L.iptr(col), L.iptr(src_rowptr), L.iptr(src_edge), L.iptr(src_dst),
               L.fptr(xl), L.fptr(xr), L.fptr(att), slope, L.fptr(edge_scale) if has_scale else None, L.fptr(alpha),
               L.fptr(grad_out), L.fptr(ws), L.fptr(gxl), L.fptr(gxr), L.fptr(gatt), L.stream_of(xl))
        return gxl, gxr, gatt, None, None, None, None, None, None


def csr_by_source(rowptr, col):
    """The by-target CSR (rowptr, col) re-indexed BY SOURCE node for the atomics-free backward (include/mgar_ops.h,
    mgar_gatv2_bwd): (src_rowptr (n+1), src_edge (E) edge ids of every node's outgoing edges, ascending, src_dst (E) their
    targets), all int32.  No host sync (graph-capturable)."""
    n = rowptr.numel() - 1
    e = col.numel()
    dst = torch.repeat_interleave(torch.arange(n, device=col.device, dtype=torch.int32), (rowptr[1:] - rowptr[:-1]).long(),
                                  output_size=e)
    order = torch.argsort(col.long(), stable=True)
    src_rowptr = torch.zeros(n + 1, dtype=torch.int32, device=col.device)
    src_rowptr[1:] = torch.cumsum(torch.bincount(col.long(), minlength=n)[:n], 0).int()
    return src_rowptr, order.int().contiguous(), dst[order].contiguous()


def edges_to_csr(edge_index: torch.Tensor, num_nodes: int, add_self_loops: bool = True):
    """(2, E) [source; target] -> CSR grouped by target (rowptr int32 (n+1), col int32 (E')).
    Self loops are removed then re-added once per node, as PyG's GATv2Conv does."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    if add_self_loops:
        keep = src != dst
        loops = torch.arange(num_nodes, device=edge_index.device)
        src = torch.cat([src[keep], loops])
        dst = torch.cat([dst[keep], loops])
    order = torch.argsort(dst, stable=True)
    col = src[order].int().contiguous()
    deg = torch.bincount(dst, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int32, device=edge_index.device)
    rowptr[1:] = torch.cumsum(deg, 0).int()
    return rowptr, col


class GATv2Conv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True,
                 negative_slope: float = 0.2, dropout: float = 0.0, add_self_loops: bool = True, bias: bool = True,
                 share_weights: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.add_self_loops, self.share_weights = add_self_loops, share_weights
        self.lin_l = nn.Linear(in_channels, heads * out_channels, bias=bias)
        self.lin_r = self.lin_l if share_weights else nn.Linear(in_channels, heads * out_channels, bias=bias)
        self.att = nn.Parameter(torch.empty(1, heads, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(heads * out_channels if concat else out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin_l.weight)
        glorot_(self.lin_r.weight)
        for lin in (self.lin_l, self.lin_r):
            if lin.bias is not None:
                nn.init.zeros_(lin.bias)
        glorot_(self.att)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor, return_attention_weights: bool = False):
        n = x.shape[0]
        xl = self.lin_l(x)
        xr = xl if self.share_weights else self.lin_r(x)
        # the CSR form depends only on the edge list: keep it on the tensor object (edge lists of fully connected
        # scenes are cached by the caller), which also keeps the boolean-mask / bincount host syncs out of the step
        cached = getattr(edge_index, "_mgar_csr", None)
        if cached is not None and cached[0] == (n, self.add_self_loops, edge_index._version):
            rowptr, col, by_source = cached[1]
        else:
            rowptr, col = edges_to_csr(edge_index, n, self.add_self_loops)
            by_source = csr_by_source(rowptr, col) if x.is_cuda else None
            try:
                edge_index._mgar_csr = ((n, self.add_self_loops, edge_index._version), (rowptr, col, by_source))
            except (AttributeError, RuntimeError):
                pass
        edge_scale = None
        if self.training and self.dropout > 0:
            keep = 1.0 - self.dropout
            edge_scale = torch.bernoulli(torch.full((col.numel(), self.heads), keep, device=x.device)) / keep
        # node features of one actor graph are A x 4096 floats: the edge-softmax / aggregate kernel stays fp32 on every
        # configuration (bf16 configurations: lin_l / lin_r run as bf16 GEMMs under autocast and are widened here)
        out, alpha = _GatAggregate.apply(xl.float(), xr.float(), self.att.view(self.heads, self.out_channels).float(), rowptr, col,
                                         edge_scale, self.heads, self.negative_slope, by_source)
        out = out if self.concat else out.view(n, self.heads, self.out_channels).mean(dim=1)
        if self.bias is not None:
            out = out + self.bias
        return (out, (rowptr, col, alpha)) if return_attention_weights else out
