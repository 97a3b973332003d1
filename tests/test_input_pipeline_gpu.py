"""Input preparation on the GPU (csrc/input_prep.hip through the C ABI) against Pillow, the oracle and the host loader."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import jrdb_tree

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pil_resize.npz")


def _pil(img, oh, ow):
    return np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))


def test_resized_bytes_equal_committed_pillow_outputs():
    from multimodal_gar_amd import input_ops
    g = np.load(GOLDEN)
    k = 0
    while "in_%d" % k in g:
        want = g["out_%d" % k]
        got = input_ops.resize_bytes(torch.from_numpy(g["in_%d" % k]).cuda().unsqueeze(0), want.shape[:2])
        assert np.array_equal(got[0].cpu().numpy(), want), "case %d" % k
        k += 1


@pytest.mark.parametrize("size", [
    (480, 3760, 720, 1280),      # the JRDB stitched frame -> the shipped image_size (Multimodal_cfg/mil3.yaml:25)
    (480, 3760, 480, 3760),      # nothing to do: a copy
    (480, 752, 224, 224), (37, 53, 11, 7), (10, 10, 10, 30), (33, 21, 33, 64), (5, 7, 50, 3), (100, 3, 7, 3), (9, 200, 9, 13),
    (129, 257, 131, 129),        # tile edges: 129 columns = one full tile + one column
    (700, 40, 35, 300),          # 20x down vertically (one output row per workgroup), 7.5x up horizontally
    (1, 1, 5, 5)])
def test_resized_bytes_equal_pillow(size):
    from multimodal_gar_amd import input_ops
    ih, iw, oh, ow = size
    rng = np.random.default_rng(sum(size))
    frames = rng.integers(0, 256, (2, ih, iw, 3), dtype=np.uint8)
    frames[1] = np.clip(np.add.outer(np.arange(ih) * 3, np.arange(iw))[..., None] % 300, 0, 255)    # smooth, saturating
    got = input_ops.resize_bytes(torch.from_numpy(frames).cuda(), (oh, ow)).cpu().numpy()
    for f in range(2):
        assert np.array_equal(got[f], _pil(frames[f], oh, ow)), "frame %d" % f


@pytest.mark.parametrize("frames,ih,iw,oh,ow", [(5, 1, 1, 2, 3), (1, 1, 1, 4, 4), (3, 7, 5, 9, 11), (4, 3, 3, 3, 3), (2, 31, 211, 17, 130)])
def test_frames_whose_size_is_not_a_multiple_of_four_bytes(frames, ih, iw, oh, ow):
    """Frame starts that are not word aligned, the last words of the buffer, sources of a few bytes."""
    from multimodal_gar_amd import input_ops
    src = np.random.default_rng(frames * 100 + iw).integers(0, 256, (frames, ih, iw, 3), dtype=np.uint8)
    got = input_ops.resize_bytes(torch.from_numpy(src).cuda(), (oh, ow)).cpu().numpy()
    for f in range(frames):
        assert np.array_equal(got[f], _pil(src[f], oh, ow)), "frame %d" % f


def test_too_strong_downscale_is_refused():
    from multimodal_gar_amd import _lib, input_ops
    with pytest.raises(_lib.MgarError):
        input_ops.resize_bytes(torch.zeros((1, 4000, 8, 3), dtype=torch.uint8, device="cuda"), (2, 8))
    with pytest.raises(_lib.MgarError):
        input_ops.resize_bytes(torch.zeros((1, 8, 8, 3), dtype=torch.uint8), (4, 4))           # host tensor: no CPU path
    assert input_ops.resize_bytes(torch.zeros((0, 8, 8, 3), dtype=torch.uint8, device="cuda"), (4, 4)).shape == (0, 4, 4, 3)


@pytest.mark.parametrize("layout", ["tchw", "cthw"])
def test_normalised_clip_equals_oracle(oracle, layout):
    from multimodal_gar_amd import input_ops
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, (5, 48, 376, 3), dtype=np.uint8)
    want = np.stack([oracle.to_tensor_normalize(oracle.pil_bilinear_resize(f, 72, 128)) for f in frames])       # (T, 3, H, W)
    if layout == "cthw":
        want = want.transpose(1, 0, 2, 3)
    dev = torch.from_numpy(frames).cuda()
    got = input_ops.resize_normalize(dev, (72, 128), layout=layout)
    assert got.dtype == torch.float32 and np.array_equal(got.cpu().numpy(), want)                  # bit for bit
    # into one clip of a batch tensor; bf16 = the float32 result rounded once
    batch = torch.full((3,) + got.shape, 9.0, device="cuda")
    assert input_ops.resize_normalize(dev, (72, 128), layout=layout, out=batch[1]).data_ptr() == batch[1].data_ptr()
    assert torch.equal(batch[1], got) and (batch[0] == 9).all() and (batch[2] == 9).all()
    half = input_ops.resize_normalize(dev, (72, 128), layout=layout, dtype=torch.bfloat16)
    assert torch.equal(half, got.to(torch.bfloat16))
    other = input_ops.resize_normalize(dev, (72, 128), mean=(0.5, 0.4, 0.3), std=(0.2, 0.3, 0.4), layout=layout)
    w2 = np.stack([oracle.to_tensor_normalize(oracle.pil_bilinear_resize(f, 72, 128), (0.5, 0.4, 0.3), (0.2, 0.3, 0.4)) for f in frames])
    assert np.array_equal(other.cpu().numpy(), w2 if layout == "tchw" else w2.transpose(1, 0, 2, 3))


@pytest.mark.parametrize("sizes", [(700, 500, 4), (0, 300, 4), (300, 0, 5), (0, 0, 4), (150000, 131072, 4), (1025, 1023, 3)])
def test_velodyne_merge_crop_equals_oracle(oracle, sizes):
    from multimodal_gar_amd import input_ops
    from multimodal_gar_amd.data.utils import jrdb_transforms as jt
    nu, nl, c = sizes
    rng = np.random.default_rng(nu + nl + c)
    up = (rng.normal(size=(nu, c)) * 40).astype(np.float32)
    lo = (rng.normal(size=(nl, c)) * 40).astype(np.float32)
    tu, tl = jt.rigid_transform("upper"), jt.rigid_transform("lower")
    lim = [-50, -30, -5, 45, 60, 5]
    if nu:
        up[0, :2] = [-50 - tu[0, 3], 0]                            # a point near the inclusive bound
    want = oracle.velodyne_merge_crop(up, lo, tu, tl, lim)
    got = input_ops.velodyne_merge_crop(torch.from_numpy(up).cuda(), torch.from_numpy(lo).cuda(), tu, tl, lim)
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want)
    if nu + nl:
        none = input_ops.velodyne_merge_crop(torch.from_numpy(up).cuda(), torch.from_numpy(lo).cuda(), tu, tl, [500, 500, 0, 600, 600, 0])
        assert none.shape == (0, c)
        every = input_ops.velodyne_merge_crop(torch.from_numpy(up).cuda(), torch.from_numpy(lo).cuda(), tu, tl,
                                              [-np.inf, -np.inf, 0, np.inf, np.inf, 0])
        assert every.shape == (nu + nl, c)


def test_device_clip_prep_equals_the_host_loader(tmp_path):
    """The 12-tuple made on the GPU from raw samples against the one the host loader (the reference's route) makes."""
    from multimodal_gar_amd.dataloader import DeviceClipPrep, JRDB_act
    root, _ = jrdb_tree.make_tree(tmp_path, missing=(("clark-center", 7),))
    cfg = jrdb_tree.loader_config()
    host = JRDB_act(cfg, root, True, jrdb_tree.NUM_ACTIONS, False)
    raw = JRDB_act(cfg, root, True, jrdb_tree.NUM_ACTIONS, False, device_prep=True)
    ids = [0, 5, 6]
    want = host.collate_batch([host[i] for i in ids])
    for layout in ("tchw", "cthw"):
        got = DeviceClipPrep(raw, layout=layout)(raw.collate_batch([raw[i] for i in ids]))
        assert len(got) == 12 and got[0].is_cuda
        rgb = got[0] if layout == "tchw" else got[0].permute(0, 2, 1, 3, 4)
        assert torch.equal(rgb.cpu(), want[0])
        for k in (1, 3, 4, 5, 6, 7, 8, 9, 10):
            assert torch.equal(got[k].cpu(), want[k]), k
        assert got[2] == want[2]
        for key in ("points", "voxels", "voxel_coords", "voxel_num_points", "gt_boxes"):
            assert np.array_equal(got[11][key].cpu().numpy(), np.asarray(want[11][key], dtype=np.float32)), key
        assert got[11]["batch_size"] == 3


def test_device_clip_prep_random_steps(tmp_path):
    """Sub-sampling to num_points and the shuffle are random: the result is a permutation / subset with the right counts."""
    from multimodal_gar_amd.dataloader import DeviceClipPrep, JRDB_act
    root, _ = jrdb_tree.make_tree(tmp_path)
    for num_points in (400, 3000):
        cfg = jrdb_tree.loader_config(num_points=num_points, shuffle=True)
        raw = JRDB_act(cfg, root, True, jrdb_tree.NUM_ACTIONS, False, device_prep=True)
        gen = torch.Generator(device="cuda").manual_seed(3)
        out = DeviceClipPrep(raw, generator=gen)(raw.collate_batch([raw[1]]))[11]
        full = JRDB_act(jrdb_tree.loader_config(), root, True, jrdb_tree.NUM_ACTIONS, False)[1][-1]
        pts, allp = out["points"][:, 1:].cpu().numpy(), np.asarray(full["points"])
        rows = set(map(tuple, allp))
        assert set(map(tuple, pts)) <= rows
        if num_points == 400:
            assert len(pts) <= 400 and len(np.unique(pts, axis=0)) == len(pts)
        else:
            assert len(np.unique(pts, axis=0)) == len(allp) and len(pts) > len(allp)
        assert int(out["voxel_num_points"].sum().item()) <= len(pts)


def test_loader_batch_drives_the_model(tmp_path):
    """A batch from the device path goes through GAR_Fusion_ALL.forward as train_func.py:111 would pass it."""
    from multimodal_gar_amd import workload as W
    from multimodal_gar_amd.dataloader import DeviceClipPrep, JRDB_act
    from multimodal_gar_amd.model.gat_model import GAR_Fusion_ALL
    root, _ = jrdb_tree.make_tree(tmp_path, n_upper=1400, n_lower=1000)
    cfg = jrdb_tree.loader_config(image_size=(64, 96), num_frames=5, num_points=2048)
    cfg.DATA_PROCESSOR = cfg.DATA_PROCESSOR[:2]                   # PointNet2MSG route: points only
    cfg.POINT_CLOUD_RANGE = [-12.0, -12.0, -2.0, 12.0, 12.0, 2.0]  # wide enough to keep every sampled point
    raw = JRDB_act(cfg, root, True, jrdb_tree.NUM_ACTIONS, False, device_prep=True)
    batch = DeviceClipPrep(raw)(raw.collate_batch([raw[raw.frames.index((0, 6))]]))
    assert batch[11]["points"].shape == (2048, 5)
    mcfg = W.model_cfg(4, 2048, gat=True, route="pointnet2")
    mcfg.DATALOADER.train.augmentation.num_boxes = 6
    net = GAR_Fusion_ALL(mcfg, W.SyntheticDataset()).cuda().eval()
    with torch.no_grad():
        out = net(batch)
    assert len(out) == 16 and all(torch.isfinite(o).all() for o in out if torch.is_tensor(o) and o.is_floating_point())
