#!/usr/bin/env python3
"""Times individual HIP kernels of libmgar_hip.so at BASELINE config c3 shapes (events on the launch
stream).  Also the command run under ``rocprofv3 --pmc ...`` to collect HBM traffic per kernel:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/microbench.py --only fps
Usage: python tools/microbench.py [--frames 120] [--only fps,ball,nn,interp,bn,qg,dw,pw] [--iters 5]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--only", default="fps,ball,nn,interp,bn,qg,dw,pw")
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    only = set(a.only.split(","))
    from multimodal_gar_amd import synthetic as S, bn_ops, nn_utils
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as pb
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as ps
    f, n = a.frames, a.points
    sc = S.scene_batch(3, f, 32, n)
    xyz = torch.from_numpy(np.ascontiguousarray(sc["points"][:, :, :3])).cuda()
    m = n // 4
    rows = []

    def rec(name, ms, byts=None, pairs=None):
        rows.append((name, ms, byts, pairs))
        extra = ""
        if byts:
            extra += "  %.0f GB/s (%.1f%% of 8 TB/s)" % (byts / ms / 1e6, byts / ms / 1e6 / 80)
        if pairs:
            extra += "  %.2e pairs/s" % (pairs / ms * 1e3)
        print("%-58s %8.3f ms%s" % (name, ms, extra), flush=True)

    if "fps" in only:
        for nn_, mm in ((n, m), (n // 4, n // 16), (n // 16, n // 64)):
            x = xyz[:, :nn_].contiguous()
            rec("fps N=%d -> M=%d (%d clouds)" % (nn_, mm, f), timeit(lambda: pb.farthest_point_sample(x, mm), a.iters),
                f * (20 * nn_ + 4 * mm), f * (mm - 1) * nn_)
    idx = pb.farthest_point_sample(xyz, m)
    new_xyz = torch.gather(xyz, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
    if "ball" in only:
        for r, ns in ((0.1, 16), (0.5, 32)):
            rec("ball_query r=%.1f ns=%d M=%d x N=%d" % (r, ns, m, n), timeit(lambda: pb.ball_query(r, ns, xyz, new_xyz), a.iters),
                f * (12 * n + 12 * m + 4 * m * ns), f * m * n)
    if "nn" in only:
        rec("three_nn n=%d x m=%d" % (n, m), timeit(lambda: pb.three_nn(xyz, new_xyz), a.iters), f * (12 * n + 12 * m + 24 * n),
            f * n * m)
    if "interp" in only:
        dist, i3 = pb.three_nn(xyz, new_xyz)
        w = (1.0 / (dist + 1e-8)); w = (w / w.sum(2, keepdim=True)).contiguous()
        c = 256
        feats = torch.randn(f, c, m, device="cuda", requires_grad=True)
        out = pb.three_interpolate(feats, i3, w)
        g = torch.randn_like(out)
        rec("three_interpolate fwd c=%d" % c, timeit(lambda: pb.three_interpolate(feats, i3, w), a.iters), f * (24 * n + 4 * c * n + 4 * c * m))
        rec("three_interpolate bwd c=%d" % c, timeit(lambda: torch.autograd.grad(out, feats, g, retain_graph=True), a.iters),
            f * (24 * n + 4 * c * n + 4 * c * m))
    if "bn" in only:
        x = torch.randn(f, 32, m, 32, device="cuda", requires_grad=True)      # SA level 1, scale 2, layer 1
        bn = torch.nn.BatchNorm2d(32).cuda().train()
        el = x.numel()
        rec("bn_act fwd (stats+apply) %s" % (tuple(x.shape),), timeit(lambda: bn_ops.bn_act(x, bn, True), a.iters), el * 12)
        y = bn_ops.bn_act(x, bn, True); g = torch.randn_like(y)
        rec("bn_act bwd (reduce+apply)", timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True), a.iters), el * 20)
        rec("bn_act_maxpool fwd (stats+max)", timeit(lambda: bn_ops.bn_act_maxpool(x, bn, True), a.iters), el * 8)
        y = bn_ops.bn_act_maxpool(x, bn, True); g = torch.randn_like(y)
        rec("bn_act_maxpool bwd", timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True), a.iters), el * 8)
    if "qg" in only:
        mq = f * 32 * 216
        cnt = torch.full((f,), n, dtype=torch.int32, device="cuda"); qcnt = torch.full((f,), 32 * 216, dtype=torch.int32, device="cuda")
        sx = xyz.reshape(-1, 3).contiguous()
        b3 = torch.from_numpy(sc["bboxes3d"][:, :32]).cuda()
        from multimodal_gar_amd.pcdet.models.roi_heads.voxelrcnn_head import global_grid_points_of_roi
        q = global_grid_points_of_roi(b3, 6)[0].reshape(-1, 3).contiguous()
        zf = torch.randn(f * n, 32, device="cuda", requires_grad=True)
        wx = torch.randn(32, 3, device="cuda")
        fn = lambda: ps._FusedQueryGroupProj.apply(0.8, 16, sx, cnt, q, qcnt, zf, wx)  # noqa: E731
        ms_ = mq * 16
        rec("ball_query(stack)+qg_proj fwd M=%d ns=16 C=32" % mq, timeit(fn, a.iters), ms_ * 4 * (1 + 3 + 32))
        y, _ = fn(); g = torch.randn_like(y)
        rec("qg_proj bwd (scatter + dWx)", timeit(lambda: torch.autograd.grad(y, zf, g, retain_graph=True), a.iters), ms_ * 4 * 32 * 2)
    if "dw" in only:
        x = torch.randn(f, 32, m * 32, device="cuda"); dy = torch.randn(f, 64, m * 32, device="cuda")
        rec("pointwise_dw Cin=32 Cout=64 P=%d" % (m * 32), timeit(lambda: nn_utils.pointwise_dw(x, dy), a.iters), x.numel() * 4 + dy.numel() * 4)
        del x, dy
        for cin, cout, cols in ((64, 128, 1024 * 32), (96, 128, 1024 * 32), (128, 196, 256 * 32), (196, 256, 256 * 32), (64, 64, 1024 * 16)):
            x = torch.randn(f, cin, cols, device="cuda"); dy = torch.randn(f, cout, cols, device="cuda")
            rec("pointwise_dw Cin=%d Cout=%d P=%d" % (cin, cout, cols), timeit(lambda: nn_utils.pointwise_dw(x, dy), a.iters), x.numel() * 4 + dy.numel() * 4)
            rec("   library  sum_b dy[b] x[b]^T", timeit(lambda: torch.einsum("bop,bip->oi", dy, x), a.iters), x.numel() * 4 + dy.numel() * 4)
            del x, dy
    if "pw" in only:
        from multimodal_gar_amd import _lib as L
        for cin, cout, ns in ((16, 16, 16), (16, 32, 16), (32, 32, 32), (32, 64, 32), (64, 64, 16)):
            p = m * ns
            x = torch.randn(f, cin, p, device="cuda"); w = torch.randn(cout, cin, device="cuda")
            y = torch.empty(f, cout, p, device="cuda")
            mean = torch.zeros(cin, device="cuda"); invstd = torch.ones(cin, device="cuda")
            fn = lambda: L.call("mgar_pointwise_conv_fwd", L.fptr(x), f, cin, p, L.fptr(w), cin, 1, cout, L.fptr(mean),  # noqa: E731
                                L.fptr(invstd), None, None, 1, L.fptr(y), L.stream_of(x))
            rec("pointwise_conv_fwd bn+relu %d->%d P=%d" % (cin, cout, p), timeit(fn, a.iters), (x.numel() + y.numel()) * 4)
            we = w.unsqueeze(0).expand(f, -1, -1)
            rec("   library bmm %d->%d (no activation)" % (cin, cout), timeit(lambda: torch.bmm(we, x, out=y), a.iters), (x.numel() + y.numel()) * 4)
            # the layer's data gradient through the same kernel: dX = W^T dY, identity activation
            dy = y; dx = x
            fn2 = lambda: L.call("mgar_pointwise_conv_fwd", L.fptr(dy), f, cout, p, L.fptr(w), 1, cin, cin, None, None, None, None, 0,  # noqa: E731
                                 L.fptr(dx), L.stream_of(dy))
            rec("   dX = W^T dY %d->%d" % (cout, cin), timeit(fn2, a.iters), (x.numel() + y.numel()) * 4)
            del x, y


if __name__ == "__main__":
    main()
