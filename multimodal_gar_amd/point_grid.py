"""Uniform cell grid over the points of a set of clouds (csrc/ball_query_grid.hip): built once per cloud set, shared by the
ball queries of every radius over it.  Device only; the grid is NEVER cached across calls -- inside a captured HIP graph a
grid built outside the capture would go stale when the static inputs are overwritten -- callers that query several radii
build it once and pass it on."""
import torch

from . import _lib as L

MIN_POINTS_PER_CLOUD = 2048   # below this the scan kernels are as fast (and the grid's four launches are not free)
MAX_NSAMPLE = 64              # csrc/ball_query_grid.hip keeps the row in registers
ENABLED = True


class PointGrid:
    """ws: the library's workspace (geometry, cell starts, points in cell order); layout: (B, n_batch | 0, n_total)."""

    def __init__(self, xyz, cell, xyz_batch_cnt=None):
        assert xyz.is_cuda and xyz.dtype == torch.float32 and xyz.is_contiguous()
        if xyz_batch_cnt is None:
            self.B, self.n_batch = xyz.shape[0], xyz.shape[1]
            self.n_total = self.B * self.n_batch
        else:
            self.B, self.n_batch, self.n_total = xyz_batch_cnt.shape[0], 0, xyz.shape[0]
        nbytes = L.raw("mgar_point_grid_workspace_bytes", self.B, self.n_total)
        self.ws = torch.empty(((nbytes + 15) // 16 * 4,), dtype=torch.float32, device=xyz.device)
        L.call("mgar_point_grid_build", self.B, self.n_batch, self.n_total, L.fptr(xyz),
               L.iptr(xyz_batch_cnt) if xyz_batch_cnt is not None else None, float(cell), L.fptr(self.ws), L.stream_of(xyz))


def wanted(points_per_cloud, nsamples):
    return ENABLED and points_per_cloud >= MIN_POINTS_PER_CLOUD and max(nsamples) <= MAX_NSAMPLE


def cell_for(radii):
    """Cell edge for a set of query radii: 3/4 of the largest, but not below the smallest (measured at the RoI lift's radii
    0.4 / 0.8 / 1.6: cell 0.4 -> 2.30 ms, 0.8 -> 1.80, 1.2 -> 1.67 for the three queries of config c3; the library enlarges
    the cell until the grid fits its cell budget, which is what decides for the trunk's small radii)."""
    rs = [float(r) for r in radii]
    return max(min(rs), 0.75 * max(rs), 1e-3)
