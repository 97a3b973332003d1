"""CPU tests: the C oracle against independent brute-force definitions (tests/bruteforce.py)
and against the committed golden fixtures.  No GPU, no product code."""
import os

import numpy as np
import pytest

import bruteforce as bf

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def lattice(rng, n, lo=-3, hi=4):
    return rng.integers(lo, hi, size=(n, 3)).astype(np.float32)


@pytest.mark.parametrize("n,m,ns,radius", [(64, 16, 4, 2.0), (300, 40, 8, 1.5), (257, 33, 16, 3.0)])
def test_ball_query_batch_lattice_exact(oracle, n, m, ns, radius):
    rng = np.random.default_rng(n)
    xyz = np.stack([lattice(rng, n), lattice(rng, n)])
    new_xyz = np.stack([xyz[0, rng.choice(n, m, replace=False)], lattice(rng, m, -6, 7)])
    idx = oracle.ball_query_batch(radius, ns, xyz, new_xyz)
    for b in range(2):
        rows, _ = bf.ball_query_rows(new_xyz[b], xyz[b], radius, ns)
        for i, r in enumerate(rows):
            if r is None:
                assert (idx[b, i] == 0).all()  # untouched (caller's zeros)
            else:
                np.testing.assert_array_equal(idx[b, i], r)


def test_ball_query_batch_random_band(oracle):
    rng = np.random.default_rng(7)
    xyz = rng.uniform(-2, 2, (1, 2000, 3)).astype(np.float32)
    new_xyz = xyz[:, :200].copy()
    idx = oracle.ball_query_batch(0.4, 16, xyz, new_xyz)
    rows, amb = bf.ball_query_rows(new_xyz[0], xyz[0], 0.4, 16, band=1e-5)
    checked = 0
    for i, r in enumerate(rows):
        if amb[i] or r is None:
            continue
        np.testing.assert_array_equal(idx[0, i], r)
        checked += 1
    assert checked > 150


def test_ball_query_stack_segments_and_empty(oracle):
    rng = np.random.default_rng(3)
    cnt = np.array([50, 0, 77], np.int32)
    qcnt = np.array([10, 5, 12], np.int32)
    xyz = lattice(rng, int(cnt.sum()))
    new_xyz = lattice(rng, int(qcnt.sum()), -8, 9)
    idx = oracle.ball_query_stack(2.0, 6, xyz, cnt, new_xyz, qcnt)
    ps = np.concatenate([[0], np.cumsum(cnt)]); qs = np.concatenate([[0], np.cumsum(qcnt)])
    for b in range(3):
        rows, _ = bf.ball_query_rows(new_xyz[qs[b]:qs[b + 1]], xyz[ps[b]:ps[b + 1]], 2.0, 6)
        for i, r in enumerate(rows):
            got = idx[qs[b] + i]
            if r is None:
                assert got[0] == -1 and (got[1:] == 0).all()
            else:
                np.testing.assert_array_equal(got, r)  # LOCAL indices


@pytest.mark.parametrize("n,m", [(64, 20), (100, 37), (1024, 64), (1500, 50), (40, 40), (2, 2), (1, 1)])
def test_fps_batch_tie_rule(oracle, n, m):
    """Integer lattice => many exact ties; the closed-form priority must reproduce the
    simulated block_size-strided scan + reduction tree."""
    rng = np.random.default_rng(n * 31 + m)
    p = lattice(rng, n, -2, 3)
    idx, temp = oracle.fps_batch(p[None], m)
    bs = oracle.opt_n_threads(n)
    ref, rtemp = bf.fps_lattice(p, m, bs)
    np.testing.assert_array_equal(idx[0], ref)
    np.testing.assert_array_equal(temp[0], rtemp)


def test_fps_block_size_rule(oracle):
    # pointnet2_batch/src/cuda_utils.h:10-14 -- includes the floating-point log quirk
    for n, want in [(1, 1), (2, 2), (3, 2), (1000, 512), (1024, 1024), (16384, 1024), (65536, 1024)]:
        assert oracle.opt_n_threads(n) == want


def test_fps_stack_global_indices(oracle):
    rng = np.random.default_rng(11)
    cnt = np.array([130, 1100, 64], np.int32)
    npnt = np.array([10, 33, 64], np.int32)
    p = lattice(rng, int(cnt.sum()), -2, 3)
    idx, _ = oracle.fps_stack(p, cnt, npnt)
    ps = np.concatenate([[0], np.cumsum(cnt)]); os_ = np.concatenate([[0], np.cumsum(npnt)])
    for b in range(3):
        ref, _ = bf.fps_lattice(p[ps[b]:ps[b + 1]], int(npnt[b]), 1024)  # stack kernel: always 1024 threads
        np.testing.assert_array_equal(idx[os_[b]:os_[b + 1]], ref + ps[b])


@pytest.mark.parametrize("n,m", [(50, 30), (100, 2), (10, 1), (64, 3)])
def test_three_nn_lattice(oracle, n, m):
    rng = np.random.default_rng(n + m)
    u = lattice(rng, n); k = lattice(rng, m)
    d2, idx = oracle.three_nn_batch(u[None], k[None])
    rd, ri = bf.three_nn_lattice(u, k)
    np.testing.assert_array_equal(d2[0], rd)          # +inf for missing neighbours
    np.testing.assert_array_equal(idx[0][:, :min(3, m)], ri[:, :min(3, m)])
    if m < 3:
        assert (idx[0][:, m:] == 0).all()


def test_three_nn_stack_global_idx(oracle):
    rng = np.random.default_rng(5)
    ucnt = np.array([20, 31], np.int32); kcnt = np.array([9, 14], np.int32)
    u = lattice(rng, 51); k = lattice(rng, 23)
    d2, idx = oracle.three_nn_stack(u, ucnt, k, kcnt)
    rd0, ri0 = bf.three_nn_lattice(u[:20], k[:9]); rd1, ri1 = bf.three_nn_lattice(u[20:], k[9:])
    np.testing.assert_array_equal(d2, np.concatenate([rd0, rd1]))
    np.testing.assert_array_equal(idx, np.concatenate([ri0, ri1 + 9]))


def test_gather_group_interpolate_against_numpy(oracle):
    rng = np.random.default_rng(9)
    b, c, n, m, ns = 2, 5, 40, 7, 3
    pts = rng.standard_normal((b, c, n)).astype(np.float32)
    gi = rng.integers(0, n, (b, m)).astype(np.int32)
    np.testing.assert_array_equal(oracle.gather_points(pts, gi), np.take_along_axis(pts, gi[:, None, :].repeat(c, 1), 2))
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    ref = np.stack([pts[bi][:, idx[bi]] for bi in range(b)])
    np.testing.assert_array_equal(oracle.group_points_batch(pts, idx), ref)
    g = rng.standard_normal((b, c, m, ns)).astype(np.float32)
    want = np.zeros((b, c, n), np.float64)
    for bi in range(b):
        for ci in range(c):
            np.add.at(want[bi, ci], idx[bi].ravel(), g[bi, ci].ravel().astype(np.float64))
    np.testing.assert_allclose(oracle.group_points_grad_batch(g, idx, n), want, rtol=1e-5, atol=1e-6)
    # interpolation
    i3 = rng.integers(0, n, (b, m, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (b, m, 3)).astype(np.float32)
    out = oracle.three_interpolate_batch(pts, i3, w)
    ref = np.stack([(pts[bi][:, i3[bi]].astype(np.float64) * w[bi][None]).sum(-1) for bi in range(b)])
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)


def test_group_stack_against_numpy(oracle):
    rng = np.random.default_rng(13)
    fcnt = np.array([11, 17], np.int32); icnt = np.array([4, 6], np.int32)
    C, ns = 6, 5
    feats = rng.standard_normal((28, C)).astype(np.float32)
    idx = np.concatenate([rng.integers(0, 11, (4, ns)), rng.integers(0, 17, (6, ns))]).astype(np.int32)
    out = oracle.group_points_stack(feats, fcnt, idx, icnt)
    start = np.array([0] * 4 + [11] * 6)
    ref = np.stack([feats[start[m] + idx[m]].T for m in range(10)])
    np.testing.assert_array_equal(out, ref)


def test_voxel_query_order_and_nonstrict(oracle):
    """Dense 4x4x4 grid of unit voxels, one point per voxel at its centre: the visiting order
    is dz,dy,dx ascending and d2 == r2 is ACCEPTED (voxel_query_gpu.cu:65)."""
    R = 4
    zz, yy, xx = np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij")
    coords = np.stack([zz.ravel(), yy.ravel(), xx.ravel()], 1)
    xyz = (coords[:, ::-1] + 0.5).astype(np.float32)        # voxel centres, xyz order
    pidx = np.arange(R ** 3, dtype=np.int32).reshape(1, R, R, R)
    q = np.array([[1.5, 1.5, 1.5]], np.float32)              # centre of voxel (z1,y1,x1)
    nc = np.array([[0, 1, 1, 1]], np.int32)
    idx = oracle.voxel_query((1, 1, 1), 1.0, 8, xyz, q, nc, pidx)
    # accepted: the 6 face neighbours at d2 == 1 (non-strict) and the centre; order dz,dy,dx
    want = [pidx[0, 0, 1, 1], pidx[0, 1, 0, 1], pidx[0, 1, 1, 0], pidx[0, 1, 1, 1], pidx[0, 1, 1, 2],
            pidx[0, 1, 2, 1], pidx[0, 2, 1, 1]]
    np.testing.assert_array_equal(idx[0], np.array(want + [want[0]], np.int32))
    # empty neighbourhood
    pidx2 = np.full((1, R, R, R), -1, np.int32)
    assert oracle.voxel_query((1, 1, 1), 1.0, 4, xyz, q, nc, pidx2)[0, 0] == -1


def test_golden_fixtures_match_oracle(oracle):
    """The committed fixtures were produced by tests/golden/make_golden.py from this oracle;
    this guards the oracle (and the fixture files) against drift."""
    path = os.path.join(GOLD, "pointnet2_small.npz")
    g = np.load(path)
    np.testing.assert_array_equal(oracle.ball_query_batch(float(g["bq_radius"]), int(g["bq_nsample"]), g["xyz"],
                                                          g["new_xyz"]), g["bq_idx"])
    idx, temp = oracle.fps_batch(g["xyz"], int(g["fps_m"]))
    np.testing.assert_array_equal(idx, g["fps_idx"])
    d2, i3 = oracle.three_nn_batch(g["xyz"], g["new_xyz"])
    np.testing.assert_array_equal(i3, g["nn_idx"])
    np.testing.assert_array_equal(d2, g["nn_d2"])
    np.testing.assert_array_equal(
        oracle.voxel_query(tuple(g["vq_range"]), float(g["vq_radius"]), int(g["vq_nsample"]), g["vq_xyz"],
                           g["vq_new_xyz"], g["vq_new_coords"], g["vq_pidx"]), g["vq_idx"])
