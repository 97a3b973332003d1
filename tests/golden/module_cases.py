"""Seeded inputs and constructor arguments of the module-level fixtures (tests/golden/reference_pcdet_modules.npz):
shared by the generator (make_reference_module_golden.py, which runs the REFERENCE's module classes) and by the
tests (which run this repo's classes on the same inputs).  Data only -- no reference code."""
import numpy as np
import torch


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _cloud(rng, b, n, extent=4.0):
    xyz = rng.uniform(-extent, extent, (b, n, 3)).astype(np.float32)
    xyz[..., 2] *= 0.25
    xyz[:, 7] = xyz[:, 3]          # duplicate points: tie rules
    return xyz


def _batch_sa(seed):
    rng = np.random.default_rng(seed)
    xyz = _cloud(rng, 2, 384)
    feats = rng.standard_normal((2, 6, 384)).astype(np.float32)
    return [(_t(xyz), False), (_t(feats), True)]


def _batch_fp(seed):
    rng = np.random.default_rng(seed)
    unknown = _cloud(rng, 2, 300)
    known = unknown[:, ::5][:, :48].copy() + 0.01
    uf = rng.standard_normal((2, 5, 300)).astype(np.float32)
    kf = rng.standard_normal((2, 9, 48)).astype(np.float32)
    return [(_t(unknown), False), (_t(known), False), (_t(uf), True), (_t(kf), True)]


def _stack_sa(seed):
    rng = np.random.default_rng(seed)
    cnt = np.array([260, 200], np.int32)
    xyz = np.concatenate([_cloud(rng, 1, 260)[0], _cloud(rng, 1, 200)[0]])
    new_xyz = np.concatenate([xyz[:40] + 0.02, xyz[260:300] - 0.03, np.array([[99., 99., 99.]], np.float32)]).astype(np.float32)
    ncnt = np.array([40, 41], np.int32)                 # the last query of sample 1 has an empty ball
    feats = rng.standard_normal((460, 7)).astype(np.float32)
    return [(_t(xyz), False), (_t(cnt), False), (_t(new_xyz), False), (_t(ncnt), False), (_t(feats), True)]


def _stack_fp(seed):
    rng = np.random.default_rng(seed)
    ucnt = np.array([150, 120], np.int32)
    unknown = np.concatenate([_cloud(rng, 1, 150)[0], _cloud(rng, 1, 120)[0]])
    known = np.concatenate([unknown[:30] + 0.01, unknown[150:152] + 0.01]).astype(np.float32)   # sample 1: only 2 known points
    kcnt = np.array([30, 2], np.int32)
    uf = rng.standard_normal((270, 4)).astype(np.float32)
    kf = rng.standard_normal((32, 8)).astype(np.float32)
    return [(_t(unknown), False), (_t(ucnt), False), (_t(known), False), (_t(kcnt), False), (_t(uf), True), (_t(kf), True)]


def _voxel_sa(seed):
    """Two samples of sparse voxels on a (Z, Y, X) = (6, 24, 24) grid of 0.25 x 0.25 x 0.5 m cells: voxel centres as xyz,
    the dense voxel -> row table, RoI-grid style queries (some far outside: empty), new_coords in [b, x, y, z] order as
    voxelrcnn_head.py:123-131 builds them."""
    rng = np.random.default_rng(seed)
    Z, Y, X = 6, 24, 24
    vs = np.array([0.25, 0.25, 0.5], np.float32)
    lo = np.array([-3.0, -3.0, -1.5], np.float32)
    coords, cnt = [], []
    for b in range(2):
        occ = rng.random((Z, Y, X)) < (0.18 if b == 0 else 0.10)
        z, y, x = np.nonzero(occ)
        perm = rng.permutation(len(z))
        coords.append(np.stack([np.full(len(z), b), z[perm], y[perm], x[perm]], 1))
        cnt.append(len(z))
    coords = np.concatenate(coords).astype(np.int32)           # (N, 4) [b, z, y, x]
    xyz = ((coords[:, [3, 2, 1]].astype(np.float32) + 0.5) * vs + lo).astype(np.float32)
    v2p = -np.ones((2, Z, Y, X), np.int32)
    v2p[coords[:, 0], coords[:, 1], coords[:, 2], coords[:, 3]] = np.arange(len(coords), dtype=np.int32)
    m = 90
    q = rng.uniform(-3.2, 3.2, (2, m, 3)).astype(np.float32)
    q[..., 2] = rng.uniform(-1.6, 1.6, (2, m)).astype(np.float32)
    q[:, :4] += 40.0                                             # far outside the grid: empty neighbourhoods
    new_xyz = q.reshape(-1, 3)
    ijk = np.floor((new_xyz - lo) / vs).astype(np.int32)         # (x, y, z) cell
    bidx = np.repeat(np.arange(2, dtype=np.int32), m)
    new_coords = np.concatenate([bidx[:, None], ijk], 1).astype(np.int32)    # [b, x, y, z]
    feats = rng.standard_normal((len(coords), 5)).astype(np.float32)
    return [(_t(xyz), False), (_t(np.array(cnt, np.int32)), False), (_t(new_xyz), False), (_t(np.array([m, m], np.int32)), False),
            (_t(new_coords), False), (_t(feats), True), (_t(v2p), False)]


CASES = [
    dict(name="batch_sa_msg", where="batch", cls="PointnetSAModuleMSG", seed=31, inputs=_batch_sa,
         kwargs=lambda: dict(npoint=64, radii=[0.9, 2.2], nsamples=[8, 16], mlps=[[6, 12, 16], [6, 8, 24]])),
    dict(name="batch_sa_msg_folded", where="batch", cls="PointnetSAModuleMSG", seed=32, inputs=_batch_sa,
         # first layers narrower than 3 + C: this repo's "project, then group" route on the device
         kwargs=lambda: dict(npoint=48, radii=[1.2, 2.5], nsamples=[16, 32], mlps=[[6, 4, 16], [6, 8, 8]])),
    dict(name="batch_fp", where="batch", cls="PointnetFPModule", seed=33, inputs=_batch_fp,
         kwargs=lambda: dict(mlp=[14, 16, 12])),
    dict(name="stack_sa_msg", where="stack", cls="StackSAModuleMSG", seed=34, inputs=_stack_sa,
         kwargs=lambda: dict(radii=[0.9, 2.4], nsamples=[8, 16], mlps=[[7, 16], [7, 12, 20]])),
    dict(name="stack_fp", where="stack", cls="StackPointnetFPModule", seed=35, inputs=_stack_fp,
         kwargs=lambda: dict(mlp=[12, 16])),
    dict(name="voxel_sa_msg", where="voxel", cls="NeighborVoxelSAModuleMSG", seed=36, inputs=_voxel_sa,
         kwargs=lambda: dict(query_ranges=[[2, 2, 2], [1, 3, 3]], radii=[0.6, 0.9], nsamples=[8, 16], mlps=[[5, 16, 16], [5, 12, 8]])),
]


def make_inputs(case):
    return case["inputs"](case["seed"])
