"""Seeded synthetic workload generator.

The reference ships no data and its loader depends on modules that are not in the repo
(dataloader.py:8-9), so every benchmark and test input comes from here.  numpy only; the
arrays are moved to the device by the caller.

  point clouds : per frame P points; 70 % background x,y ~ U(-20,20), z ~ U(-2,2) and 30 %
                 drawn inside the actor boxes; intensity U(0,1); 1 % exact duplicates so the
                 tie rules of FPS / three_nn / ball_query are exercised.
  actors       : A boxes per scene, (cx,cy,cz,l,w,h,rot_z) as in dataloader.py:190, padded to
                 `num_boxes` with zeros and person_id = -1 (dataloader.py:245-253).
  images       : N(0,1) float32 (B,T,3,H,W).
"""
import numpy as np


def actor_boxes3d(rng, n_actors):
    c = np.zeros((n_actors, 7), np.float32)
    c[:, 0:2] = rng.uniform(-15, 15, (n_actors, 2))
    c[:, 2] = 0.0
    c[:, 3] = rng.uniform(0.4, 1.0, n_actors)
    c[:, 4] = rng.uniform(0.4, 1.0, n_actors)
    c[:, 5] = rng.uniform(1.4, 2.0, n_actors)
    c[:, 6] = rng.uniform(-np.pi, np.pi, n_actors)
    return c


def actor_boxes2d(rng, n_actors, height, width, min_side=16):
    min_side = min(min_side, max(min(height, width) // 8, 1))   # tiny smoke-test images
    x1 = rng.uniform(0, width - 4 * min_side, n_actors)
    y1 = rng.uniform(0, height - 4 * min_side, n_actors)
    w = rng.uniform(min_side, np.minimum(width - x1, 12 * min_side))
    h = rng.uniform(min_side, np.minimum(height - y1, 20 * min_side))
    return np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)


def point_cloud(rng, n_points, boxes3d, frac_in_box=0.3, dup_frac=0.01):
    """(n_points, 4) float32 [x, y, z, intensity]."""
    n_box = int(n_points * frac_in_box) if len(boxes3d) else 0
    n_bg = n_points - n_box
    bg = np.empty((n_bg, 3), np.float32)
    bg[:, 0:2] = rng.uniform(-20, 20, (n_bg, 2))
    bg[:, 2] = rng.uniform(-2, 2, n_bg)
    parts = [bg]
    if n_box:
        which = rng.integers(0, len(boxes3d), n_box)
        b = boxes3d[which]
        local = rng.uniform(-0.5, 0.5, (n_box, 3)).astype(np.float32) * b[:, 3:6]
        cosa, sina = np.cos(b[:, 6]), np.sin(b[:, 6])
        x = local[:, 0] * cosa - local[:, 1] * sina + b[:, 0]
        y = local[:, 0] * sina + local[:, 1] * cosa + b[:, 1]
        z = local[:, 2] + b[:, 2]
        parts.append(np.stack([x, y, z], 1).astype(np.float32))
    xyz = np.concatenate(parts, 0)
    xyz = xyz[rng.permutation(n_points)]
    n_dup = int(n_points * dup_frac)
    if n_dup:
        dst = rng.choice(n_points, n_dup, replace=False)
        src = rng.integers(0, n_points, n_dup)
        xyz[dst] = xyz[src]
    inten = rng.uniform(0, 1, (n_points, 1)).astype(np.float32)
    return np.concatenate([xyz, inten], 1).astype(np.float32)


def scene_batch(seed, n_scenes, n_actors, n_points, num_boxes=None, height=720, width=1280):
    """A batch of independent frame-scenes.

    Returns dict of numpy arrays:
      points     (n_scenes, n_points, 4)
      bboxes3d   (n_scenes, num_boxes, 7), bboxes (n_scenes, num_boxes, 4) zero padded
      person_id  (n_scenes, num_boxes) int64, -1 in the pad slots (>= 1 pad slot is kept so
                 the reference's  len(unique(person_id)) - 1  actor count holds,
                 model/gat_model.py:1047)
    """
    rng = np.random.default_rng(seed)
    num_boxes = num_boxes or (n_actors + 1)
    assert num_boxes > n_actors
    pts = np.empty((n_scenes, n_points, 4), np.float32)
    b3 = np.zeros((n_scenes, num_boxes, 7), np.float32)
    b2 = np.zeros((n_scenes, num_boxes, 4), np.float32)
    pid = -np.ones((n_scenes, num_boxes), np.int64)
    for s in range(n_scenes):
        boxes = actor_boxes3d(rng, n_actors)
        b3[s, :n_actors] = boxes
        b2[s, :n_actors] = actor_boxes2d(rng, n_actors, height, width)
        pid[s, :n_actors] = np.arange(n_actors)
        pts[s] = point_cloud(rng, n_points, boxes)
    return {"points": pts, "bboxes3d": b3, "bboxes": b2, "person_id": pid}


def scene_labels(seed, n_scenes, n_actors, num_boxes=None):
    """Synthetic annotations of the shape the reference's dataloader yields (dataloader.py:245-253, 293): social_group_id
    (n_scenes, num_boxes) int64 with -1 padding (2..A/3 groups per scene), action and social_group_activity
    (n_scenes, num_boxes, 27) float32 multi-hot rows (one pose of each of the three pose groups, a few interactions)."""
    rng = np.random.default_rng(seed)
    num_boxes = num_boxes or (n_actors + 1)
    gid = -np.ones((n_scenes, num_boxes), np.int64)
    act = np.zeros((2, n_scenes, num_boxes, 27), np.float32)
    for s in range(n_scenes):
        n_groups = int(rng.integers(2, max(3, n_actors // 3 + 1)))
        g = rng.integers(0, n_groups, n_actors)
        g[:n_groups] = np.arange(n_groups)                       # every group id occurs
        gid[s, :n_actors] = g
    for a in act:
        rows = a[:, :n_actors].reshape(-1, 27)
        n = rows.shape[0]
        rows[np.arange(n), rng.integers(0, 3, n)] = 1.0
        rows[np.arange(n), 3 + rng.integers(0, 3, n)] = 1.0
        rows[np.arange(n), 6 + rng.integers(0, 4, n)] = 1.0
        rows[:, 11:25] = (rng.random((n, 14)) < 0.15).astype(np.float32)
        a[:, :n_actors] = rows.reshape(n_scenes, n_actors, 27)
    return {"social_group_id": gid, "action": act[0], "social_group_activity": act[1]}


def images(seed, n_clips, n_frames, height, width):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n_clips, n_frames, 3, height, width), dtype=np.float32)


def voxelize(xyz, voxel_size, pc_range):
    """Occupied voxels of one cloud: returns (coords (V,3) int32 [z,y,x], centres (V,3) float32,
    inverse (n,) voxel id per point).  Centres follow pcdet/utils/common_utils.py:66-82."""
    vs = np.asarray(voxel_size, np.float32)
    lo = np.asarray(pc_range[:3], np.float32)
    hi = np.asarray(pc_range[3:], np.float32)
    grid = np.round((hi - lo) / vs).astype(np.int64)  # (X, Y, Z)
    c = np.floor((xyz[:, :3] - lo) / vs).astype(np.int64)
    ok = ((c >= 0) & (c < grid)).all(1)
    c = c[ok]
    key = (c[:, 2] * grid[1] + c[:, 1]) * grid[0] + c[:, 0]
    uniq, inv = np.unique(key, return_inverse=True)
    z = uniq // (grid[1] * grid[0]); r = uniq % (grid[1] * grid[0]); y = r // grid[0]; x = r % grid[0]
    coords = np.stack([z, y, x], 1).astype(np.int32)
    centres = ((coords[:, ::-1].astype(np.float32) + 0.5) * vs + lo).astype(np.float32)
    return coords, centres, inv, ok, grid[::-1].astype(np.int32)  # grid as (Z, Y, X)
