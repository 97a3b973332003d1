"""Module-level integration parity (SURVEY.md section 4 plan): the same host-side module is run on the GPU
(HIP kernels) and on the CPU with the C oracle standing in for the kernels
(oracle/cpu_backend.py), from identical parameters and inputs.  Covers rows a8, a9, a12, a14, a15,
a16 and the whole clip model (a24) of SURVEY.md section 8."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from param_fill import fill_deterministic  # noqa: E402

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    assert a.shape == b.shape
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    from conftest import record_error
    record_error("", err, scale, rtol)
    assert err <= atol + rtol * scale, "max err %g vs scale %g" % (err, scale)


def run_both(make_module, inputs, train=True, seed=3):
    """-> (gpu_out, gpu_grads), (cpu_out, cpu_grads): outputs and input gradients from both backends."""
    from oracle.cpu_backend import use_cpu_oracle
    mod = fill_deterministic(make_module(), seed=seed)
    mod.train(train)
    res = []
    for dev in ("cuda", "cpu"):
        m = copy.deepcopy(mod).to(dev)
        ins = [t.detach().clone().to(dev).requires_grad_(t.is_floating_point() and t.dim() > 1 and rg)
               for t, rg in inputs]
        ctx = use_cpu_oracle() if dev == "cpu" else None
        if ctx:
            ctx.__enter__()
        try:
            out = m(*ins)
            outs = [o for o in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(o) and o.is_floating_point()]
            loss = sum((o * o).sum() for o in outs if o.requires_grad)
            loss.backward()
        finally:
            if ctx:
                ctx.__exit__(None, None, None)
        grads = [t.grad for t in ins if t.requires_grad] + [p.grad for p in m.parameters() if p.grad is not None]
        res.append((outs, grads))
    return res


def scene(seed, b, n):
    from multimodal_gar_amd import synthetic as S
    sc = S.scene_batch(seed, b, 4, n)
    return torch.from_numpy(np.ascontiguousarray(sc["points"][:, :, :3])), torch.from_numpy(sc["bboxes3d"])


def test_sa_msg_and_fp_modules_batch():
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_modules as M
    xyz, _ = scene(1, 2, 1024)
    feats = torch.randn(2, 5, 1024)
    (go, gg), (co, cg) = run_both(lambda: M.PointnetSAModuleMSG(npoint=128, radii=[0.8, 2.0], nsamples=[8, 16],
                                                                mlps=[[5, 16, 32], [5, 16, 32]]),
                                  [(xyz, False), (feats, True)])
    for a, b in zip(go, co):
        close(a, b)
    for a, b in zip(gg, cg):
        close(a, b, rtol=1e-4)
    known = xyz[:, :100].contiguous(); kf = torch.randn(2, 12, 100); uf = torch.randn(2, 7, 1024)
    (go, gg), (co, cg) = run_both(lambda: M.PointnetFPModule(mlp=[19, 32, 16]), [(xyz, False), (known, False), (uf, True), (kf, True)])
    for a, b in zip(go + gg, co + cg):
        close(a, b, rtol=1e-4)


def test_stack_sa_and_fp_modules():
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as M
    xyz, _ = scene(2, 2, 700)
    xyz = xyz.reshape(-1, 3)
    cnt = torch.tensor([700, 700], dtype=torch.int32)
    new_xyz = torch.cat([xyz[:50], xyz[700:760] + 0.05, torch.tensor([[900., 900., 900.]])])   # last query: empty ball
    ncnt = torch.tensor([50, 61], dtype=torch.int32)
    feats = torch.randn(1400, 9)
    (go, gg), (co, cg) = run_both(lambda: M.StackSAModuleMSG(radii=[0.9, 2.5], nsamples=[8, 16], mlps=[[9, 16], [9, 24]]),
                                  [(xyz, False), (cnt, False), (new_xyz, False), (ncnt, False), (feats, True)])
    close(go[1], co[1])
    for a, b in zip(gg, cg):
        close(a, b, rtol=1e-4)
    kf = torch.randn(111, 6)
    (go, gg), (co, cg) = run_both(lambda: M.StackPointnetFPModule(mlp=[15, 20]),
                                  [(xyz, False), (cnt, False), (new_xyz, False), (ncnt, False), (feats, True), (kf, True)])
    for a, b in zip(go + gg, co + cg):
        close(a, b, rtol=1e-4)


def test_voxel_rcnn_route_detector():
    """MeanVFE -> trunk stand-in -> VoxelRCNNHead (voxel query + NeighborVoxelSAModuleMSG)."""
    from multimodal_gar_amd import workload as W
    from multimodal_gar_amd.pcdet.models import build_network
    from oracle.cpu_backend import use_cpu_oracle
    ds = W.SyntheticDataset()
    pts, b3 = scene(3, 2, 4096)
    points = torch.cat([pts, torch.rand(2, 4096, 1)], -1)
    net = fill_deterministic(build_network(W.lidar_model_cfg(4096, "voxel"), 1, ds), seed=9).eval()
    outs = []
    for dev in ("cuda", "cpu"):
        m = copy.deepcopy(net).to(dev)
        data = W.voxelize_batch(points.to(dev), ds)
        data["gt_boxes"] = b3[:, :4, :].contiguous().to(dev)
        if dev == "cpu":
            with use_cpu_oracle(), torch.no_grad():
                outs.append(m(data)["pooled_features"])
        else:
            with torch.no_grad():
                outs.append(m(data)["pooled_features"])
    assert outs[0].shape == (8, 216, 96)
    close(outs[0], outs[1])
    assert outs[1].abs().sum() > 0


def test_clip_model_forward_gpu_vs_cpu_backend():
    from multimodal_gar_amd import workload as W
    from oracle.cpu_backend import use_cpu_oracle
    torch.manual_seed(0)
    model = fill_deterministic(W.ClipModel(4, 1024), seed=11).eval()
    batch = W.make_batch(5, 1, 2, 4, 1024, 64, 96, torch.device("cpu"))
    with use_cpu_oracle(), torch.no_grad():
        want = model(batch)
    gm = copy.deepcopy(model).cuda()
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        got = gm(gb)
    assert len(got) == 16
    for a, b in zip(got, want):
        close(a, b, rtol=1e-4, atol=1e-5)


def test_clip_model_train_mode_several_clips_gpu_vs_cpu_backend():
    """Train-mode forward of 3 clips: on the device all clips go through I3D in ONE pass with per-clip BatchNorm
    statistics on a side stream, FPS is pruned (2 048 points) and issued first; the CPU run (oracle backend) takes the
    reference-shaped route, one clip at a time.  Dropout off; BatchNorm uses batch statistics on both sides."""
    from multimodal_gar_amd import workload as W
    from oracle.cpu_backend import use_cpu_oracle
    torch.manual_seed(0)
    model = fill_deterministic(W.ClipModel(4, 2048), seed=12).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0
    batch = W.make_batch(6, 3, 2, 4, 2048, 64, 96, torch.device("cpu"))
    with use_cpu_oracle(), torch.no_grad():
        want = copy.deepcopy(model)(batch)
    gm = copy.deepcopy(model).cuda()
    assert gm.batch_i3d and gm.overlap_branches
    gb = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    with torch.no_grad():
        got = gm(gb)
    torch.cuda.synchronize()
    assert len(got) == 16
    for a, b in zip(got, want):
        close(a, b, rtol=2e-4, atol=1e-5)      # train-mode BatchNorm over few samples amplifies fp32 rounding: measured <= 6.4e-5
    # the whole coordinate-only part of the trunk (every level's FPS, ball queries, 3-NN weights) issued ahead on a third
    # stream: the same numbers, bit for bit (pcdet/models/backbones_3d/pointnet2_backbone.py: PointNet2MSG.geometry)
    assert gm.geometry_ahead == "fps1"
    gm.geometry_ahead = "all"
    with torch.no_grad():
        ahead = gm(gb)
    torch.cuda.synchronize()
    for a, b in zip(ahead, got):
        assert torch.equal(a, b)


def _reference_style_batch(seed, n_actors, n_points, route, ds):
    """The 12-tuple the reference's collate_batch produces (dataloader.py:296-419), batch size 1,
    with a pcdet data_dict holding NUMPY arrays (load_data_to_gpu moves them)."""
    from multimodal_gar_amd import synthetic as S, workload as W
    sc = S.scene_batch(seed, 1, n_actors, n_points, num_boxes=n_actors + 2, height=64, width=96)
    rng = np.random.default_rng(seed)
    images = torch.from_numpy(S.images(seed, 1, 5, 64, 96))
    bboxes = torch.from_numpy(sc["bboxes"]); b3 = torch.from_numpy(sc["bboxes3d"]); pid = torch.from_numpy(sc["person_id"])
    pts = sc["points"]
    if route == "pointnet2":
        data = {"batch_size": 1, "points": np.concatenate([np.zeros((n_points, 1), np.float32), pts[0]], 1),
                "gt_boxes": sc["bboxes3d"][:, :n_actors]}
    else:
        vd = W.voxelize_batch(torch.from_numpy(pts), ds)
        data = {"batch_size": 1, "voxels": vd["voxels"].numpy(), "voxel_num_points": vd["voxel_num_points"].numpy(),
                "voxel_coords": vd["voxel_coords"].numpy().astype(np.float32), "gt_boxes": sc["bboxes3d"][:, :n_actors]}
    return (images, bboxes, None, b3, None, pid, None, None, None, None, None, data)


@pytest.mark.parametrize("route", ["pointnet2", "voxel"])
def test_gar_fusion_all_forward_reference_call_shape(route):
    """GAR_Fusion_ALL.forward(batch) exactly as train_func.py:111 calls it: 12-tuple in, 16-tuple out,
    on both LiDAR routes; GPU (HIP kernels) vs CPU (oracle backend)."""
    from multimodal_gar_amd import workload as W
    from multimodal_gar_amd.model.gat_model import GAR_Fusion_ALL
    from oracle.cpu_backend import use_cpu_oracle
    ds = W.SyntheticDataset()
    cfg = W.model_cfg(4, 2048, gat=True, route=route)
    cfg.DATALOADER.train.augmentation.num_boxes = 6
    net = fill_deterministic(GAR_Fusion_ALL(cfg, ds), seed=21).eval()
    batch = _reference_style_batch(4, 4, 2048, route, ds)
    import unittest.mock as mock
    with use_cpu_oracle(), torch.no_grad(), mock.patch("multimodal_gar_amd.model.gat_model.load_data_to_gpu", lambda d: d.update(
            {k: torch.from_numpy(v).float() for k, v in d.items() if isinstance(v, np.ndarray)})):
        want = net(tuple(copy.deepcopy(batch)))
    gnet = copy.deepcopy(net).cuda()
    gb = list(copy.deepcopy(batch))
    for i in (0, 1, 3, 5):
        gb[i] = gb[i].cuda()
    with torch.no_grad():
        got = gnet(tuple(gb))
    assert len(got) == 16 and got[0].shape == (1, 6, 6) and got[-1].shape == (1, 1)
    for a, b in zip(got, want):
        close(a, b, rtol=1e-4, atol=1e-5)


def test_social_grouping_model_forward():
    from multimodal_gar_amd import workload as W
    from multimodal_gar_amd.model.sg_model import SocialGrouping_model
    from oracle.cpu_backend import use_cpu_oracle
    cfg = W.model_cfg(4, 1024, gat=True)
    net = fill_deterministic(SocialGrouping_model(cfg, N=2), seed=22).eval()
    batch = _reference_style_batch(6, 4, 1024, "pointnet2", None)
    with use_cpu_oracle(), torch.no_grad():
        want = net(tuple(copy.deepcopy(batch)))
    gnet = copy.deepcopy(net).cuda()
    gb = list(copy.deepcopy(batch))
    for i in (0, 1, 3, 5):
        gb[i] = gb[i].cuda()
    with torch.no_grad():
        got = gnet(tuple(gb))
    assert got.shape == (4, 4) and torch.allclose(torch.diagonal(got), torch.ones(4, device="cuda"))
    close(got, want, rtol=1e-4)


def test_project_then_group_path_equals_reference_chain():
    """SA modules whose first layer is narrower than its (3 + C) input take the "project, then group"
    route on the device; the CPU run (oracle backend) takes the reference-shaped chain."""
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_modules as MB
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as MS
    xyz, _ = scene(7, 2, 900)
    feats = torch.randn(2, 40, 900)
    mod = MB.PointnetSAModuleMSG(npoint=100, radii=[0.9, 2.0], nsamples=[8, 16], mlps=[[40, 16, 32], [40, 24, 24]])
    assert mod.mlps[0].first_layer_foldable(43)
    (go, gg), (co, cg) = run_both(lambda: MB.PointnetSAModuleMSG(npoint=100, radii=[0.9, 2.0], nsamples=[8, 16],
                                                                 mlps=[[40, 16, 32], [40, 24, 24]]),
                                  [(xyz, False), (feats, True)])
    for a, b in zip(go, co):
        close(a, b)
    for a, b in zip(gg, cg):
        close(a, b, rtol=1e-4)
    sx = xyz.reshape(-1, 3)
    cnt = torch.tensor([900, 900], dtype=torch.int32)
    new_xyz = torch.cat([sx[:70], sx[900:990] + 0.05, torch.tensor([[500., 500., 500.]])])
    ncnt = torch.tensor([70, 91], dtype=torch.int32)
    sf = torch.randn(1800, 40)
    (go, gg), (co, cg) = run_both(lambda: MS.StackSAModuleMSG(radii=[0.9, 2.5], nsamples=[8, 16], mlps=[[40, 16], [40, 24, 32]]),
                                  [(sx, False), (cnt, False), (new_xyz, False), (ncnt, False), (sf, True)])
    close(go[1], co[1])
    for a, b in zip(gg, cg):
        close(a, b, rtol=1e-4)


def test_stack_sa_msg_channel_major_features_equal_stacked_rows():
    """StackSAModuleMSG fed (B, C, n) channel-major features (what the PointNet++ trunk produces) must give what
    it gives for the stacked (B*n, C) rows: outputs, input gradient (in the other layout) and parameter gradients."""
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as MS
    torch.manual_seed(4)
    xyz, _ = scene(9, 3, 800)
    sx = xyz.reshape(-1, 3).cuda()
    cnt = torch.tensor([800, 800, 800], dtype=torch.int32, device="cuda")
    new_xyz = torch.cat([sx[:60], sx[800:900] + 0.03, sx[1600:1650] - 0.02]).contiguous()
    ncnt = torch.tensor([60, 100, 50], dtype=torch.int32, device="cuda")
    fcm = torch.randn(3, 40, 800, device="cuda")
    mod = fill_deterministic(MS.StackSAModuleMSG(radii=[0.9, 2.5, 1.5], nsamples=[8, 16, 16],
                                                 mlps=[[40, 16], [40, 24, 32], [40, 32]]), seed=8).cuda().train()
    outs, fgrads, pgrads = [], [], []
    for cm in (True, False):
        m = copy.deepcopy(mod)
        f = (fcm.clone() if cm else fcm.permute(0, 2, 1).reshape(-1, 40).contiguous()).requires_grad_(True)
        _, y = m(sx, cnt, new_xyz, ncnt, f)
        assert y.shape == (210, 16 + 32 + 32)
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view(y.shape)).sum().backward()
        outs.append(y)
        fgrads.append(f.grad if cm else f.grad.view(3, 800, 40).permute(0, 2, 1))
        pgrads.append([p.grad for p in m.parameters()])
    close(outs[0], outs[1])
    close(fgrads[0], fgrads[1], rtol=1e-4)
    for a, b in zip(*pgrads):
        close(a, b, rtol=1e-4)


def test_train_step_hip_graph_matches_eager():
    """TrainStep.capture(): forward + backward replayed from a HIP graph must give the loss and the gradients of
    the eager step from the same state (dropout off: the graph-safe RNG draws different masks), and the graph-mode
    step must keep training (parameters move, loss stays finite)."""
    from multimodal_gar_amd import workload as W
    dev = torch.device("cuda")
    batch = W.make_batch(3, 1, 2, 4, 2048, 96, 160, dev)

    def build():
        step = W.TrainStep(4, 2048, dev, seed=11, manual_allreduce=True)
        for m in step.module.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
                m.dropout = 0.0
        return step
    eager, graph = build(), build()
    graph.capture(batch, warmup=2)                        # two eager warm-up steps inside, then the capture
    assert graph.graph is not None
    eager.module.load_state_dict(graph.module.state_dict())

    def grads_of(step):
        return {n: p.grad.detach().clone() for n, p in step.module.named_parameters() if p.grad is not None}

    def worst_diff(a, b):
        return max(((a[n] - b[n]).abs().max().item() / (a[n].abs().max().item() + 1e-12)) for n in a)
    le = eager._forward_backward(batch); g1 = grads_of(eager)
    eager._forward_backward(batch); g2 = grads_of(eager)          # run-to-run noise of the eager step itself
    graph.graph.replay()
    torch.cuda.synchronize()
    gg = grads_of(graph)
    lg = graph._loss
    assert abs(float(le) - float(lg)) <= 1e-3 * abs(float(le)) + 1e-6, (float(le), float(lg))
    assert set(gg) == set(g1) and len(g1) > 100
    noise, diff = worst_diff(g1, g2), worst_diff(g1, gg)
    # float atomics reorder sums between runs and this tiny configuration amplifies that through its BatchNorms:
    # the graph must be no further from an eager run than a few times what two eager runs are from each other
    assert diff <= max(10 * noise, 1e-4), (diff, noise)
    graph.run_eager(batch)                                 # an eager step in between must not detach .grad from the graph
    before = [p.detach().clone() for p in graph.params[:5]]
    losses = [float(graph.run(batch)) for _ in range(3)]
    assert all(p.grad is g for p, g in zip(graph.params, graph._graph_grads))
    assert all(l == l and l < 1e6 for l in losses)
    assert any(not torch.equal(a, b) for a, b in zip(before, graph.params[:5]))
