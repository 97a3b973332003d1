"""Micro-benchmark of the atomic-free stacked grouping backward at the RoI-grid-lift shape of config c3
(120 samples x 1024 source points, 6912 queries per sample, nsample 16, C 32).  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel times; prints wall times per phase otherwise."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as P  # noqa: E402


def main():
    for npts in (8192, 1024):
        case(npts)


def case(npts):
    B, nq, ns, C = 120, 6912, 16, 32
    print("== %d samples x %d source rows, %d queries x %d slots, C %d" % (B, npts, nq, ns, C))
    rng = np.random.default_rng(0)
    M = B * nq
    # ball-query-like rows: `found` distinct neighbours near a per-query centre, the rest filled with the first one
    found = rng.integers(1, ns + 1, (M, 1))
    centre = rng.integers(0, npts, (M, 1))
    nbr = (centre + rng.integers(-40, 41, (M, ns))) % npts
    slot = np.arange(ns)[None, :]
    idx = np.where(slot < found, nbr, nbr[:, :1]).astype(np.int32)
    idx[::97, 0] = -1
    d_idx = torch.from_numpy(idx).cuda()
    qc = torch.full((B,), nq, dtype=torch.int32, device="cuda")
    pc = torch.full((B,), npts, dtype=torch.int32, device="cuda")
    g_t = torch.randn(M * ns, C, device="cuda")
    g = g_t.t().contiguous()
    xyz = torch.randn(B * npts, 3, device="cuda")
    new_xyz = torch.randn(M, 3, device="cuda")
    out = torch.zeros(B * npts, C, device="cuda")

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    index = P.query_group_inverse_index(B, M, ns, B * npts, d_idx, qc, pc)
    print("index build      %.3f ms" % timed(lambda: P.query_group_inverse_index(B, M, ns, B * npts, d_idx, qc, pc)))
    print("rows (+ d wx)    %.3f ms" % timed(lambda: P.query_group_proj_grad_rows_wrapper(B, M, C, ns, g_t, d_idx, qc, pc, out, xyz=xyz,
                                                                                       new_xyz=new_xyz, index=index)))
    print("atomic scatter   %.3f ms" % timed(lambda: P.query_group_proj_grad_wrapper(B, M, C, ns, g, d_idx, qc, pc, out)))
    gb = M * ns * (4 + 4 * C) / 1e9
    print("algorithmic bytes per launch: %.3f GB" % gb)


if __name__ == "__main__":
    main()
