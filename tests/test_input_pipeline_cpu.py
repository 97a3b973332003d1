"""Input pipeline on the CPU: the oracle's restatement of Pillow's resize against Pillow itself and
the committed fixture, the library's host-side coefficient helper, the PCD reader, the pcdet processor mirrors and the
``JRDB_act`` loader (reference dataloader.py) on a synthetic directory tree."""
import os
import struct

import numpy as np
import pytest
import torch
from PIL import Image

import jrdb_tree

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pil_resize.npz")


def test_oracle_resize_equals_committed_pillow_outputs(oracle):
    g = np.load(GOLDEN)
    k = 0
    while "in_%d" % k in g:
        want = g["out_%d" % k]
        got = oracle.pil_bilinear_resize(g["in_%d" % k], want.shape[0], want.shape[1])
        assert np.array_equal(got, want), "case %d" % k
        k += 1
    assert k == 7


@pytest.mark.parametrize("size", [(48, 376, 72, 128), (37, 53, 11, 7), (10, 10, 10, 30), (33, 21, 33, 64), (5, 7, 50, 3),
                                  (64, 64, 64, 64), (100, 3, 7, 3), (9, 200, 9, 13), (240, 940, 360, 320)])
def test_oracle_resize_equals_installed_pillow(oracle, size):
    ih, iw, oh, ow = size
    img = np.random.default_rng(sum(size)).integers(0, 256, (ih, iw, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(oracle.pil_bilinear_resize(img, oh, ow), want)


def test_oracle_normalisation_is_the_float32_formula(oracle):
    img = np.random.default_rng(1).integers(0, 256, (6, 9, 3), dtype=np.uint8)
    x = torch.from_numpy(img).permute(2, 0, 1).float().div(255)
    want = (x - torch.tensor(oracle.IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(oracle.IMAGENET_STD).view(3, 1, 1)
    assert np.array_equal(oracle.to_tensor_normalize(img), want.numpy())
    from multimodal_gar_amd.dataloader import resize_to_tensor_normalize
    got = resize_to_tensor_normalize(Image.fromarray(img), (6, 9))
    assert torch.equal(got, want)
    got = resize_to_tensor_normalize(Image.fromarray(img), (11, 4))
    assert np.array_equal(got.numpy(), oracle.to_tensor_normalize(oracle.pil_bilinear_resize(img, 11, 4)))


@pytest.mark.parametrize("pair", [(3760, 1280), (480, 720), (100, 100), (7, 50), (50, 7), (1, 9), (1000, 3)])
def test_library_coefficient_tables_equal_the_oracle(oracle, pair):
    """mgar_image_resample_coeffs is a HOST helper of the C ABI (no GPU involved)."""
    from multimodal_gar_amd import input_ops
    bounds, kk = input_ops.resample_tables_host(*pair)
    if pair[0] == pair[1]:
        assert kk.shape == (pair[1], 1) and (kk == 1 << 22).all() and np.array_equal(bounds[:, 0], np.arange(pair[1]))
        return
    wb, wk = oracle.resample_coeffs(*pair)
    assert np.array_equal(bounds, wb) and np.array_equal(kk, wk)


def _lzf_literal(data):
    out = bytearray()
    for i in range(0, len(data), 32):
        chunk = data[i:i + 32]
        out.append(len(chunk) - 1)
        out += chunk
    return bytes(out)


def test_pcd_reader(tmp_path):
    from multimodal_gar_amd.data.utils import utils as U
    pts = np.random.default_rng(0).normal(size=(257, 4)).astype(np.float32)
    for kind in ("binary", "ascii"):
        U.write_pcd(str(tmp_path / "a.pcd"), pts, data=kind)
        assert np.array_equal(U.load_pointcloud(str(tmp_path / "a.pcd")), pts)
    # binary_compressed: field-major body behind an LZF stream
    body = pts.T.copy().tobytes()
    comp = _lzf_literal(body)
    head = ("VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 257\nHEIGHT 1\nPOINTS 257\n"
            "DATA binary_compressed\n").encode()
    (tmp_path / "c.pcd").write_bytes(head + struct.pack("<II", len(comp), len(body)) + comp)
    assert np.array_equal(U.load_pointcloud(str(tmp_path / "c.pcd")), pts)
    # an LZF stream with back references: "abcabcabcabc" + a run of one byte
    assert U._lzf_decompress(bytes([2]) + b"abc" + bytes([(7 << 5) | 0, 0, 2]) + bytes([0]) + b"z" + bytes([(3 << 5) | 0, 0]), 3 + 9 + 1 + 5) \
        == b"abcabcabcabc" + b"zzzzzz"
    # no intensity field, extra fields, an empty cloud
    U.write_pcd(str(tmp_path / "b.pcd"), pts[:, [0, 1, 2, 3, 3]][:, :5], fields=("x", "y", "z", "ring", "t"))
    got = U.load_pointcloud(str(tmp_path / "b.pcd"))
    assert np.array_equal(got[:, :3], pts[:, :3]) and (got[:, 3] == 0).all()
    U.write_pcd(str(tmp_path / "e.pcd"), np.zeros((0, 4), np.float32))
    assert U.load_pointcloud(str(tmp_path / "e.pcd")).shape == (0, 4)


def test_get_lidar_with_sweeps():
    from multimodal_gar_amd.data.utils.utils import get_lidar_with_sweeps
    pc = np.arange(40, dtype=np.float32).reshape(10, 4)
    np.random.seed(0)
    sub = get_lidar_with_sweeps(pc, 4)
    assert sub.shape == (4, 4) and (np.diff(sub[:, 0]) > 0).all()
    more = get_lidar_with_sweeps(pc, 25)
    assert more.shape == (25, 4) and np.array_equal(more[:10], pc) and set(map(tuple, more[10:])) <= set(map(tuple, pc))
    assert get_lidar_with_sweeps(pc, -1) is pc and get_lidar_with_sweeps(pc, 10) is pc


def test_velodyne_transforms_and_oracle_merge(oracle):
    from multimodal_gar_amd.data.utils import jrdb_transforms as jt
    rng = np.random.default_rng(3)
    up, lo = rng.normal(size=(50, 4)).astype(np.float32) * 6, rng.normal(size=(70, 4)).astype(np.float32) * 6
    tu, tl = jt.rigid_transform("upper"), jt.rigid_transform("lower")
    assert tu.shape == (3, 4) and tu[2, 3] == np.float32(jt.UPPER_OFFSET[2]) and np.allclose(tu[:2, :2] @ tu[:2, :2].T, np.eye(2), atol=1e-6)
    assert np.array_equal(tl[:, :3], np.eye(3, dtype=np.float32))
    rngs = [-8, -8, -2, 8, 8, 2]
    got = oracle.velodyne_merge_crop(up, lo, tu, tl, rngs)
    a = np.concatenate([jt.transform_pts_upper_velodyne_to_base(up[:, :3].T).T, up[:, 3:]], 1)
    b = np.concatenate([jt.transform_pts_lower_velodyne_to_base(lo[:, :3].T).T, lo[:, 3:]], 1)
    allp = np.concatenate([a, b], 0)
    keep = (np.abs(allp[:, 0]) <= 8) & (np.abs(allp[:, 1]) <= 8)
    assert np.array_equal(got, allp[keep]) and 0 < keep.sum() < len(allp)
    # against float64 matrix algebra
    ref = (tu[:, :3].astype(np.float64) @ up[:, :3].T.astype(np.float64)).T + tu[:, 3]
    assert np.allclose(a[:, :3], ref, atol=1e-5)
    jt.set_calibration("lower", 0.5, (1, 2, 3))
    try:
        assert np.allclose(jt.rigid_transform("lower")[:, 3], [1, 2, 3]) and jt.rigid_transform("lower")[0, 1] < 0
    finally:
        jt.set_calibration("lower", jt.LOWER_YAW, jt.LOWER_OFFSET)


def test_data_processor_steps(oracle):
    from multimodal_gar_amd.pcdet.datasets.processor.data_processor import DataProcessor
    from multimodal_gar_amd.pcdet.datasets.processor.point_feature_encoder import PointFeatureEncoder
    cfg = jrdb_tree.loader_config(shuffle=True)
    enc = PointFeatureEncoder(cfg.POINT_FEATURE_ENCODING)
    assert enc.num_point_features == 4
    proc = DataProcessor(cfg.DATA_PROCESSOR, np.array(cfg.POINT_CLOUD_RANGE, np.float32), training=True, num_point_features=4)
    assert list(proc.grid_size) == [32, 32, 4]
    rng = np.random.default_rng(5)
    pts = np.concatenate([rng.uniform(-10, 10, (900, 2)), rng.uniform(-1.9, 1.9, (900, 1)), rng.uniform(0, 1, (900, 1))], 1).astype(np.float32)
    boxes = np.array([[0, 0, 0, 1, 1, 1, 0.3], [9, 0, 0, 1, 1, 1, 0], [0, 0, 5, 1, 1, 1, 0]], np.float32)
    d = enc.forward({"points": pts.copy(), "gt_boxes": boxes})
    np.random.seed(11)
    d = proc.forward(d)
    inside = pts[(np.abs(pts[:, 0]) <= 8) & (np.abs(pts[:, 1]) <= 8)]
    np.random.seed(11)
    shuffled = inside[np.random.permutation(len(inside))]
    assert np.array_equal(d["points"], shuffled) and len(d["gt_boxes"]) == 1
    v, c, n = oracle.voxelize_points_loop(shuffled, [0.5, 0.5, 1.0], cfg.POINT_CLOUD_RANGE, 4, 400)
    assert np.array_equal(d["voxels"], v) and np.array_equal(d["voxel_coords"], c) and np.array_equal(d["voxel_num_points"], n)
    # sample_points: all far points kept, exact count
    cfg2 = jrdb_tree.loader_config()
    cfg2.DATA_PROCESSOR = [{"NAME": "sample_points", "NUM_POINTS": {"train": 300, "test": -1}}]
    p2 = DataProcessor(cfg2.DATA_PROCESSOR, np.array([-100, -100, -5, 100, 100, 5], np.float32), True, 4)
    far = pts.copy()
    far[:50, 0] += 60
    out = p2.forward({"points": far})["points"]
    assert out.shape == (300, 4) and (np.linalg.norm(out[:, :3], axis=1) >= 40).sum() == (np.linalg.norm(far[:, :3], axis=1) >= 40).sum()
    grown = p2.forward({"points": far[:200]})["points"]         # fewer points than asked: every point once, the rest repeats
    assert grown.shape == (300, 4) and len(np.unique(grown, axis=0)) == 200
    with pytest.raises(ValueError):                               # more repeats than points: numpy refuses, as in the reference
        p2.forward({"points": far[:100]})


def test_jrdb_act_clip_against_hand_computation(tmp_path, oracle):
    from multimodal_gar_amd.dataloader import JRDB_act
    from multimodal_gar_amd.data.utils import jrdb_transforms as jt
    from multimodal_gar_amd.data.utils.utils import load_pointcloud
    root, anns = jrdb_tree.make_tree(tmp_path, missing=(("clark-center", 7),))
    cfg = jrdb_tree.loader_config()
    ds = JRDB_act(cfg, root, True, jrdb_tree.NUM_ACTIONS, False)
    assert len(ds) == 8 and ds.frames[0] == (0, 4) and ds.get_frames((1, 6)) == [(1, 6, 5), (1, 6, 6), (1, 6, 7)]
    idx = ds.frames.index((1, 6))
    images, bboxes, src_fid, b3, bnum, pid, gid, seq_id, frame_id, actions, gact, dd = ds[idx]
    base = os.path.join(root, "train_dataset_with_activity")
    want = []
    for fid in (5, 6, 6):                                         # frame 7 is missing: the key frame stands in
        jpg = np.asarray(Image.open(os.path.join(base, "images/image_stitched/clark-center/%06d.jpg" % fid)).convert("RGB"))
        want.append(oracle.to_tensor_normalize(oracle.pil_bilinear_resize(jpg, 36, 64)))
    assert images.shape == (3, 3, 36, 64) and np.array_equal(images.numpy(), np.stack(want))
    ann = anns[1][6]
    k = len(ann["person_id"])
    x, y, w, h = ann["bboxes_2d"][0]
    assert np.allclose(bboxes[0].numpy(), [x * 64, y * 36, (x + w) * 64, (y + h) * 36]) and (bboxes[k:] == 0).all() and bboxes.shape == (6, 4)
    assert src_fid == 6 and b3.shape == (6, 7) and np.isclose(b3[0, 6].item(), ann["bboxes_3d"][0]["rot_z"]) and (b3[k:] == 0).all()
    assert bnum.tolist() == [k] * 3 and pid.tolist() == ann["person_id"] + [-1] * (6 - k) and gid.tolist() == ann["social_group_id"] + [-1] * (6 - k)
    assert seq_id.shape == (3, 6) and (seq_id[:, :k] == 1).all() and (seq_id[:, k:] == -1).all()
    assert frame_id[:, 0].tolist() == [5, 6, 7] and (frame_id[:, k:] == -1).all()
    assert actions.shape == (6, 5) and actions[:k].tolist() == [[float(v) for v in a] for a in ann["actions"]] and (actions[k:] == 0).all()
    assert gact[:k].tolist() == [[float(v) for v in a[::-1]] for a in ann["actions"]]
    # LiDAR: both sensors of the KEY frame, base frame, range crop, voxels
    up = load_pointcloud(os.path.join(base, "pointclouds/upper_velodyne/clark-center/000006.pcd"))
    lo = load_pointcloud(os.path.join(base, "pointclouds/lower_velodyne/clark-center/000006.pcd"))
    pts = oracle.velodyne_merge_crop(up, lo, jt.rigid_transform("upper"), jt.rigid_transform("lower"), cfg.POINT_CLOUD_RANGE)
    assert np.array_equal(np.asarray(dd["points"]), pts) and 0 < len(pts) < 1200
    v, c, n = oracle.voxelize_points_loop(pts, [0.5, 0.5, 1.0], cfg.POINT_CLOUD_RANGE, 4, 400)
    assert np.array_equal(np.asarray(dd["voxels"]), v) and np.array_equal(np.asarray(dd["voxel_coords"]), c)
    g = np.array([[b[k2] for k2 in ("cx", "cy", "cz", "l", "w", "h", "rot_z")] for b in ann["bboxes_3d"]], np.float32)
    assert np.array_equal(dd["gt_boxes"], g[(np.abs(g[:, :2]) <= 8).all(1) & (np.abs(g[:, 2]) <= 2)])
    # a key frame without an image file falls back to sample 0 (dataloader.py:160-163)
    fb = ds[ds.frames.index((1, 7))]
    assert fb[2] == 4 and (fb[7][:, 0] == 0).all()
    # literal row indexing of the reference touches three rows only
    lit = JRDB_act(cfg, root, True, jrdb_tree.NUM_ACTIONS, False, literal_transform_rows=True)
    raw = np.concatenate([up, lo], 0)
    got = lit.load_pc(os.path.join(base, "pointclouds/lower_velodyne/clark-center/000006.pcd"))
    assert np.array_equal(got[3:len(up)], raw[3:len(up)]) and not np.array_equal(got[:3], raw[:3])
    # too many boxes: an error, not an endless loop
    few = JRDB_act(jrdb_tree.loader_config(num_boxes=0), root, True, jrdb_tree.NUM_ACTIONS, False)
    with pytest.raises(ValueError):
        few[0]
    # fine-tuning mode draws one frame of the window
    ft = JRDB_act(cfg, root, True, jrdb_tree.NUM_ACTIONS, True)
    for _ in range(5):
        (s, a, f), = ft.get_frames((0, 4))
        assert (s, a) == (0, 4) and 4 <= f <= 6


def test_collate_batch_and_torch_dataloader(tmp_path):
    from multimodal_gar_amd.dataloader import JRDB_act, RawClipBatch
    root, _ = jrdb_tree.make_tree(tmp_path)
    ds = JRDB_act(jrdb_tree.loader_config(), root, True, jrdb_tree.NUM_ACTIONS, False)
    loader = torch.utils.data.DataLoader(ds, batch_size=3, num_workers=0, collate_fn=ds.collate_batch, shuffle=False)
    batch = next(iter(loader))
    assert len(batch) == 12
    rgb, bboxes, src, b3, bnum, pid, gid, sid, fid, act, gact, ret = batch
    assert rgb.shape == (3, 3, 3, 36, 64) and bboxes.shape == (3, 6, 4) and src == [4, 5, 6] and b3.shape == (3, 6, 7)
    assert bnum.shape == (3, 3) and pid.shape == (3, 6) and pid.dtype == torch.float32 and sid.shape == (3, 3, 6) and act.shape == (3, 6, 5)
    assert ret["batch_size"] == 3 and ret["points"].shape[1] == 5 and ret["voxel_coords"].shape[1] == 4
    singles = [ds[i] for i in range(3)]
    at = 0
    for i, s in enumerate(singles):
        n = len(s[-1]["points"])
        assert (ret["points"][at:at + n, 0] == i).all() and np.array_equal(ret["points"][at:at + n, 1:], np.asarray(s[-1]["points"]))
        at += n
    assert at == len(ret["points"]) and len(ret["voxels"]) == sum(len(s[-1]["voxels"]) for s in singles) == len(ret["voxel_num_points"])
    assert ret["gt_boxes"].shape == (3, max(len(s[-1]["gt_boxes"]) for s in singles), 7) and ret["use_lead_xyz"].tolist() == [True] * 3
    # key rules of the generic collate
    out = JRDB_act.collate_pcdet([{"images": np.ones((4, 6, 3)), "points_2d": np.ones((2, 2)), "calib": "a", "gt_boxes2d": np.zeros((0, 4)),
                                   "gt_dense": 1, "frame": np.array(1)},
                                  {"images": np.ones((5, 3, 3)), "points_2d": np.ones((5, 2)), "calib": "b", "gt_boxes2d": np.ones((2, 4)),
                                   "gt_dense": 2, "frame": np.array(2)}])
    assert out["images"].shape == (2, 5, 6, 3) and out["images"][0, 4].sum() == 0 and out["images"][1, :, 3:].sum() == 0
    assert out["points_2d"].shape == (2, 5, 2) and out["calib"] == ["a", "b"] and out["gt_boxes2d"].shape == (2, 2, 4)
    assert "gt_dense" not in out and out["frame"].tolist() == [1, 2]
    # raw samples for the device path
    raw = JRDB_act(jrdb_tree.loader_config(), root, True, jrdb_tree.NUM_ACTIONS, False, device_prep=True)
    rb = raw.collate_batch([raw[0], raw[1]])
    assert isinstance(rb, RawClipBatch) and len(rb) == 2 and rb.frames_u8.shape == (2, 3, 24, 188, 3) and rb.frames_u8.dtype == torch.uint8
    assert rb.upper[0].shape == (700, 4) and rb.lower[1].shape == (500, 4) and rb.src_fid == [4, 5]


def test_loader_shards_across_two_ranks(tmp_path):
    """The clip index space splits over ranks with torch's DistributedSampler (one process per GPU, no data-path collective):
    disjoint, together complete, and every rank's batches collate."""
    from torch.utils.data.distributed import DistributedSampler
    from multimodal_gar_amd.dataloader import JRDB_act
    root, _ = jrdb_tree.make_tree(tmp_path)
    ds = JRDB_act(jrdb_tree.loader_config(), root, True, jrdb_tree.NUM_ACTIONS, False)
    seen = []
    for rank in range(2):
        sampler = DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=False)
        loader = torch.utils.data.DataLoader(ds, batch_size=2, sampler=sampler, collate_fn=ds.collate_batch)
        keys = []
        for batch in loader:
            assert batch[0].shape[1:] == (3, 3, 36, 64) and batch[11]["batch_size"] == batch[0].shape[0]
            keys += [(int(s), int(f)) for s, f in zip(batch[7][:, 0, 0].tolist(), batch[2])]
        seen.append(keys)
    assert len(seen[0]) == len(seen[1]) == 4 and not set(seen[0]) & set(seen[1])
    assert set(seen[0]) | set(seen[1]) == set(ds.frames)
