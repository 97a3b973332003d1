"""Generates tests/golden/pointnet2_small.npz from the CPU oracle on seeded synthetic inputs.

The reference ships no golden vectors and its native path cannot run here (CUDA only), so
these fixtures pin the ORACLE (and, through the GPU tests, the HIP kernels) against drift;
they are not outputs of the reference.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multimodal_gar_amd import synthetic as S  # noqa: E402  (numpy-only module)
from oracle import oracle as O  # noqa: E402


def main():
    sc = S.scene_batch(seed=2023, n_scenes=2, n_actors=4, n_points=512)
    xyz = np.ascontiguousarray(sc["points"][:, :, :3])
    fps_m = 64
    fps_idx, _ = O.fps_batch(xyz, fps_m)
    new_xyz = np.stack([xyz[b][fps_idx[b]] for b in range(2)])
    out = {"xyz": xyz, "new_xyz": new_xyz, "fps_m": fps_m, "fps_idx": fps_idx}
    out["bq_radius"], out["bq_nsample"] = np.float32(1.0), 16
    out["bq_idx"] = O.ball_query_batch(1.0, 16, xyz, new_xyz)
    out["nn_d2"], out["nn_idx"] = O.three_nn_batch(xyz, new_xyz)
    # voxel query: scene 0, 0.5 m voxels
    pc_range = [-20, -20, -2, 20, 20, 2]
    coords, centres, _, _, grid = S.voxelize(xyz[0], [0.5, 0.5, 0.5], pc_range)
    pidx = -np.ones((1, grid[0], grid[1], grid[2]), np.int32)
    pidx[0, coords[:, 0], coords[:, 1], coords[:, 2]] = np.arange(len(coords), dtype=np.int32)
    q = new_xyz[0]
    qc = np.floor((q - np.array(pc_range[:3], np.float32)) / 0.5).astype(np.int32)
    new_coords = np.concatenate([np.zeros((len(q), 1), np.int32), qc[:, ::-1]], 1)
    out.update(vq_xyz=centres, vq_new_xyz=q, vq_new_coords=new_coords, vq_pidx=pidx,
               vq_range=np.array([2, 2, 2], np.int32), vq_radius=np.float32(1.0), vq_nsample=8)
    out["vq_idx"] = O.voxel_query((2, 2, 2), 1.0, 8, centres, q, new_coords, pidx)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pointnet2_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
