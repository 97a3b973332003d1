"""Ball query timing at the shapes of config c3 (and, with --c5, c5): scan kernel (one launch per radius), multi-radius
scan, and the cell-grid kernel (csrc/ball_query_grid.hip; grid build included) -- trunk levels 1-2 (batch layout, FPS
centres: rows fill up) and the RoI-grid lift (stack layout: rows rarely fill)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd import point_grid as G, synthetic as S
from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as CB, pointnet2_utils as pb
from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as CS


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3


c5 = "--c5" in sys.argv
f, n, a = (30, 65536, 128) if c5 else (120, 16384, 32)
sc = S.scene_batch(1, f, a, n)
xyz = torch.from_numpy(sc["points"][:, :, :3].copy()).cuda().contiguous()
level = xyz
for lvl, (m, radii) in enumerate(((n // 4, (0.1, 0.5)), (n // 16, (0.5, 1.0)), (n // 64, (1.0, 2.0)))):
    ns = (16, 32)
    nn = level.shape[1]
    idx = pb.farthest_point_sample(level, m)
    new_xyz = torch.gather(level, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
    outs = [torch.zeros(f, m, k, dtype=torch.int32, device="cuda") for k in ns]

    def grid_all():
        g = G.PointGrid(level, G.cell_for(radii))
        for r, k, o in zip(radii, ns, outs):
            CB.ball_query_grid_wrapper(f, nn, m, r, k, new_xyz, g, o)
    t1 = sum(timeit(lambda r=r, k=k, o=o: CB.ball_query_scan_wrapper(f, nn, m, r, k, new_xyz, level, o)) for r, k, o in zip(radii, ns, outs))
    G.ENABLED = False
    t2 = timeit(lambda: CB.ball_query_multi_wrapper(f, nn, m, list(radii), list(ns), new_xyz, level, outs))
    G.ENABLED = True
    t3 = timeit(grid_all)
    print("batch level %d (%d -> %d) radii %s: scans %.2f ms, multi-scan %.2f ms, grid %.2f ms" % (lvl + 1, nn, m, radii, t1, t2, t3))
    level = new_xyz
# RoI grid: a boxes x 216 grid points per cloud
from multimodal_gar_amd.pcdet.models.roi_heads.voxelrcnn_head import global_grid_points_of_roi
rois = torch.from_numpy(sc["bboxes3d"][:, :a]).cuda().float()
g, _ = global_grid_points_of_roi(rois, 6)
q = g.view(-1, 3).contiguous(); M = q.shape[0]
qcnt = torch.full((f,), a * 216, dtype=torch.int32, device="cuda"); pcnt = torch.full((f,), n, dtype=torch.int32, device="cuda")
sx = xyz.view(-1, 3).contiguous()
radii, ns = (0.4, 0.8, 1.6), (16, 16, 16)
outs = [torch.zeros(M, k, dtype=torch.int32, device="cuda") for k in ns]


def grid_roi(cell):
    gr = G.PointGrid(sx, cell, pcnt)
    for r, k, o in zip(radii, ns, outs):
        CS.ball_query_grid_wrapper(f, M, r, k, q, qcnt, gr, o)


t1 = sum(timeit(lambda r=r, k=k, o=o: CS.ball_query_scan_wrapper(f, M, r, k, q, qcnt, sx, pcnt, o)) for r, k, o in zip(radii, ns, outs))
print("stack RoI radii %s: scans %.2f ms; grid: %s" % (radii, t1, ", ".join("cell %.1f -> %.2f ms" % (c, timeit(lambda c=c: grid_roi(c))) for c in (0.4, 0.6, 0.8, 1.2))))
print("grid build alone: %.3f ms" % timeit(lambda: G.PointGrid(sx, 0.4, pcnt)))
