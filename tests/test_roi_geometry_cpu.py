"""CPU tests of the pure-torch geometry around voxel RoI pooling
against independent numpy formulas."""
import numpy as np
import torch


def test_grid_points_of_roi_match_numpy():
    from multimodal_gar_amd.pcdet.models.roi_heads.voxelrcnn_head import global_grid_points_of_roi
    rng = np.random.default_rng(0)
    rois = np.concatenate([rng.uniform(-5, 5, (7, 3)), rng.uniform(0.5, 2, (7, 3)), rng.uniform(-3, 3, (7, 1))], 1).astype(np.float32)
    glob, local = global_grid_points_of_roi(torch.from_numpy(rois), 6)
    assert glob.shape == (7, 216, 3)
    # reference formula (voxelrcnn_head.py:167-188): idx from nonzero() of a (6,6,6) ones tensor = x slowest, z fastest
    ii = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    for r in range(7):
        loc = (ii + 0.5) / 6 * rois[r, 3:6] - rois[r, 3:6] / 2
        np.testing.assert_allclose(local[r].numpy(), loc, rtol=1e-5, atol=1e-6)
        c, s = np.cos(rois[r, 6]), np.sin(rois[r, 6])
        rot = np.stack([loc[:, 0] * c - loc[:, 1] * s, loc[:, 0] * s + loc[:, 1] * c, loc[:, 2]], 1)   # x -> y positive
        np.testing.assert_allclose(glob[r].numpy(), rot + rois[r, :3], rtol=1e-4, atol=1e-5)


def test_voxel_centers_voxel2pinds_meanvfe():
    from multimodal_gar_amd.pcdet.utils import common_utils as cu
    from multimodal_gar_amd.pcdet.utils.spconv_utils import SparseConvTensor as SparseTensorLite
    from multimodal_gar_amd.pcdet.models.backbones_3d.vfe import MeanVFE
    coords = torch.tensor([[0, 1, 2], [3, 0, 5]])                      # z, y, x
    centres = cu.get_voxel_centers(coords, 2, [0.1, 0.2, 0.5], [-1.0, -2.0, -3.0, 1, 2, 3])
    want = (np.array([[2, 1, 0], [5, 0, 3]]) + 0.5) * np.array([0.2, 0.4, 1.0]) + np.array([-1.0, -2.0, -3.0])
    np.testing.assert_allclose(centres.numpy(), want, rtol=1e-6)
    idx = torch.tensor([[0, 0, 1, 2], [1, 3, 0, 5], [0, 3, 3, 3]], dtype=torch.int32)
    v2p = cu.generate_voxel2pinds(SparseTensorLite(torch.zeros(3, 1), idx, [4, 4, 6], 2))
    assert v2p.shape == (2, 4, 4, 6) and v2p.dtype == torch.int32
    assert v2p[0, 0, 1, 2] == 0 and v2p[1, 3, 0, 5] == 1 and v2p[0, 3, 3, 3] == 2 and (v2p == -1).sum() == 2 * 4 * 4 * 6 - 3
    voxels = torch.tensor([[[1., 2., 3., 4.], [3., 4., 5., 6.], [0., 0., 0., 0.]], [[0., 0., 0., 0.]] * 3])
    out = MeanVFE(None, 4)({"voxels": voxels, "voxel_num_points": torch.tensor([2., 0.])})["voxel_features"]
    np.testing.assert_allclose(out.numpy(), [[2, 3, 4, 5], [0, 0, 0, 0]])
