"""csrc/ball_query_grid.hip (round 3): ball query through a uniform cell grid -- the reference's rows bit for bit (C oracle:
pointnet2_batch/src/ball_query_gpu.cu:15-51, pointnet2_stack/src/ball_query_gpu.cu:16-66), and identical to the scan kernel
it replaces for large clouds.  Edge cases: queries far outside the cloud, empty balls, balls holding more than nsample
points (the nsample SMALLEST indices must survive whatever the visiting order), duplicate points, radius 0, degenerate
clouds (one point repeated, a line), ragged stacked batches with empty samples, nsample 1 .. 64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(seed, b, n):
    from multimodal_gar_amd import synthetic as S
    return np.ascontiguousarray(S.scene_batch(seed, b, 6, n)["points"][:, :, :3])


def _grid_batch(xyz, new_xyz, radius, nsample, cell=None):
    from multimodal_gar_amd import point_grid as G
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as CB
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    x, q = dev(xyz), dev(new_xyz)
    idx = torch.zeros((b, m, nsample), dtype=torch.int32, device="cuda")
    CB.ball_query_grid_wrapper(b, n, m, radius, nsample, q, G.PointGrid(x, cell if cell else G.cell_for([radius])), idx)
    return idx.cpu().numpy()


@pytest.mark.parametrize("n,m,radius,nsample", [(2048, 300, 0.8, 16), (5000, 777, 0.4, 32), (16384, 1000, 1.6, 16), (4096, 256, 3.0, 64),
                                                 (3000, 100, 0.05, 1), (2500, 64, 0.0, 8), (8192, 500, 10.0, 5)])
def test_grid_ball_query_batch_matches_oracle(oracle, n, m, radius, nsample):
    xyz = _scene(n + m, 2, n)
    rng = np.random.default_rng(n)
    # queries: cloud points, jittered cloud points, and points far outside the cloud's box
    new_xyz = np.concatenate([xyz[:, :m - 8] + rng.normal(0, 0.05, (2, m - 8, 3)).astype(np.float32),
                              np.full((2, 4, 3), 500.0, np.float32), np.full((2, 4, 3), -1e6, np.float32)], 1)
    want = oracle.ball_query_batch(radius, nsample, xyz, new_xyz)
    np.testing.assert_array_equal(_grid_batch(xyz, new_xyz, radius, nsample), want)
    # the cell edge is a tuning knob, never a correctness one
    for cell in (0.03, 0.7, 25.0):
        np.testing.assert_array_equal(_grid_batch(xyz, new_xyz, radius, nsample, cell=cell), want)


def test_grid_ball_query_degenerate_clouds(oracle):
    rng = np.random.default_rng(2)
    one = np.tile(np.array([[1.5, -2.0, 0.25]], np.float32), (1, 2200, 1))                        # every point identical
    line = np.zeros((1, 2200, 3), np.float32); line[0, :, 0] = np.linspace(-30, 30, 2200)         # zero extent in y, z
    lattice = rng.integers(-3, 4, (1, 2200, 3)).astype(np.float32)                                # exact ties of d2 == r2
    for xyz in (one, line, lattice):
        new_xyz = np.concatenate([xyz[:, :50], xyz[:, 100:150] + 0.5], 1)
        for radius, ns in ((1.0, 16), (2.0, 32)):
            want = oracle.ball_query_batch(radius, ns, xyz, new_xyz)
            np.testing.assert_array_equal(_grid_batch(xyz, new_xyz, radius, ns), want)


def test_grid_ball_query_stack_ragged_matches_oracle(oracle):
    from multimodal_gar_amd import point_grid as G
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as CS
    sc = _scene(9, 3, 6000)
    cnt = np.array([6000, 0, 3500, 4100], np.int32)                       # an empty sample in the middle
    xyz = np.concatenate([sc[0], sc[1][:3500], sc[2][:4100]])
    qcnt = np.array([400, 30, 0, 513], np.int32)                          # queries against the empty sample; a sample without queries
    q = np.concatenate([sc[0][:400] + 0.02, sc[1][:30], sc[2][:512] - 0.03, [[900., 900., 900.]]]).astype(np.float32)
    for radius, ns in ((0.5, 16), (1.6, 16), (0.9, 40)):
        want = oracle.ball_query_stack(radius, ns, xyz, cnt, q, qcnt)
        x, qq, c, qc = dev(xyz), dev(q), dev(cnt), dev(qcnt)
        idx = torch.zeros((q.shape[0], ns), dtype=torch.int32, device="cuda")
        CS.ball_query_grid_wrapper(4, q.shape[0], radius, ns, qq, qc, G.PointGrid(x, G.cell_for([radius]), c), idx)
        np.testing.assert_array_equal(idx.cpu().numpy(), want)
        scan = torch.zeros_like(idx)
        CS.ball_query_scan_wrapper(4, q.shape[0], radius, ns, qq, qc, x, c, scan)
        assert torch.equal(idx, scan)


def test_grid_equals_scan_kernel_at_c3_roi_size():
    """The RoI-grid lift's queries at config c3's per-frame size (32 actors x 216 grid points against 16 384 points, radii
    0.4 / 0.8 / 1.6, nsample 16), 4 frames: the grid kernel's rows == the scan kernel's rows, and it is faster."""
    from multimodal_gar_amd import point_grid as G, synthetic as S
    from multimodal_gar_amd.pcdet.models.roi_heads.voxelrcnn_head import global_grid_points_of_roi
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as CS
    f, a, p = 4, 32, 16384
    sc = S.scene_batch(3, f, a, p)
    xyz = dev(sc["points"][:, :, :3].reshape(-1, 3))
    grid_xyz, _ = global_grid_points_of_roi(dev(sc["bboxes3d"][:, :a]), 6)
    q = grid_xyz.view(-1, 3).contiguous()
    cnt = torch.full((f,), p, dtype=torch.int32, device="cuda")
    qcnt = torch.full((f,), a * 216, dtype=torch.int32, device="cuda")
    outs, times = [], []
    for which in range(2):
        for rep in range(2):     # the second repetition is the timed one
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            grid = G.PointGrid(xyz, 0.4, cnt) if which == 0 else None
            rows = []
            for radius in (0.4, 0.8, 1.6):
                idx = torch.zeros((q.shape[0], 16), dtype=torch.int32, device="cuda")
                if which == 0:
                    CS.ball_query_grid_wrapper(f, q.shape[0], radius, 16, q, qcnt, grid, idx)
                else:
                    CS.ball_query_scan_wrapper(f, q.shape[0], radius, 16, q, qcnt, xyz, cnt, idx)
                rows.append(idx)
            t1.record()
        torch.cuda.synchronize()
        outs.append(rows)
        times.append(t0.elapsed_time(t1))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    print("RoI-grid ball queries, 4 frames x 3 radii: grid (incl. build) %.3f ms, scan %.3f ms" % tuple(times))
    assert times[0] < times[1]


# ---------------------------------------------------------------------------------------------------- three_nn through the grid
def _three_nn_grid_batch(unknown, known, cell=0.0):
    from multimodal_gar_amd import point_grid as G
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as CB
    b, n, _ = unknown.shape
    m = known.shape[1]
    d = torch.empty((b, n, 3), dtype=torch.float32, device="cuda")
    i = torch.empty((b, n, 3), dtype=torch.int32, device="cuda")
    CB.three_nn_grid_wrapper(b, n, m, dev(unknown), G.PointGrid(dev(known), cell), d, i)
    return d.cpu().numpy(), i.cpu().numpy()


@pytest.mark.parametrize("n,m", [(4000, 2048), (16384, 4096), (3000, 9000), (500, 2100)])
def test_grid_three_nn_batch_matches_oracle(oracle, n, m):
    sc = _scene(n + m, 2, max(n, m))
    unknown = np.ascontiguousarray(sc[:, :n])
    known = np.ascontiguousarray(sc[:, -m:] + 0.013)
    unknown[:, :6] = [[500, 500, 500], [-700, 3, 1], [0, 0, 90], [25, -25, 3], [-1e5, 0, 0], [20, 20, 2]]   # far outside the known cloud
    want_d, want_i = oracle.three_nn_batch(unknown, known)
    for cell in (0.0, -1.0, 0.3, 5.0):                       # automatic (4 / 1 point per cell) and fixed cell edges
        got_d, got_i = _three_nn_grid_batch(unknown, known, cell)
        np.testing.assert_array_equal(got_i, want_i)
        np.testing.assert_array_equal(got_d, want_d)


def test_grid_three_nn_ties_and_degenerate_clouds(oracle):
    rng = np.random.default_rng(4)
    lattice = rng.integers(-6, 7, (1, 3000, 3)).astype(np.float32)            # equal distances everywhere: the index decides
    one = np.tile(np.array([[2.0, 1.0, -1.0]], np.float32), (1, 2500, 1))
    line = np.zeros((1, 2500, 3), np.float32); line[0, :, 1] = np.linspace(-40, 40, 2500)
    for known in (lattice, one, line):
        unknown = np.concatenate([known[:, :400] + 0.5, known[:, 400:800], rng.uniform(-50, 50, (1, 200, 3)).astype(np.float32)], 1)
        want_d, want_i = oracle.three_nn_batch(unknown, known)
        got_d, got_i = _three_nn_grid_batch(unknown, known)
        np.testing.assert_array_equal(got_i, want_i)
        np.testing.assert_array_equal(got_d, want_d)


def test_grid_three_nn_stack_ragged_matches_oracle(oracle):
    from multimodal_gar_amd import point_grid as G
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as CS
    sc = _scene(21, 3, 7000)
    kcnt = np.array([5000, 2, 0, 3000], np.int32)                         # fewer than three known points; none at all
    known = np.concatenate([sc[0][:5000], sc[1][:2], sc[2][:3000]])
    ucnt = np.array([800, 50, 20, 0], np.int32)
    unknown = np.concatenate([sc[0][5000:5800], sc[1][100:150], sc[2][4000:4020]]).astype(np.float32)
    want_d, want_i = oracle.three_nn_stack(unknown, ucnt, known, kcnt)
    d = torch.empty((unknown.shape[0], 3), dtype=torch.float32, device="cuda")
    i = torch.empty((unknown.shape[0], 3), dtype=torch.int32, device="cuda")
    CS.three_nn_grid_wrapper(dev(unknown), dev(ucnt), G.PointGrid(dev(known), 0.0, dev(kcnt)), d, i)
    np.testing.assert_array_equal(i.cpu().numpy(), want_i)
    np.testing.assert_array_equal(d.cpu().numpy(), want_d)
    d2, i2 = torch.empty_like(d), torch.empty_like(i)
    CS.three_nn_scan_wrapper(dev(unknown), dev(ucnt), dev(known), dev(kcnt), d2, i2)
    assert torch.equal(i, i2) and torch.equal(d, d2)
