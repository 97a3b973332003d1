"""Parity against outputs of the REFERENCE's own torch classes (tests/golden/
reference_torch_blocks.npz, produced by tests/golden/make_reference_golden.py by importing
/root/reference/model in the build container).  Weights are re-created on both sides with
tests/golden/param_fill.py.  Tolerance: 1e-4 relative (BASELINE.json north_star)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from param_fill import fill_deterministic  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "reference_torch_blocks.npz"))
RTOL = 1e-4


def close(a, b, rtol=RTOL, atol=1e-6):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    scale = np.abs(b).max() + 1e-12
    err = np.abs(a - b).max()
    assert err <= atol + rtol * scale, "max err %g vs scale %g" % (err, scale)


# ------------------------------------------------------------------ CPU: pure-torch blocks
@pytest.mark.parametrize("tag,cin,cint,dim", [("nl2d", 32, 4, 2), ("nl3d", 24, 3, 3)])
def test_nlblock_matches_reference(tag, cin, cint, dim):
    from multimodal_gar_amd.model.backbone import NLBlockND
    m = fill_deterministic(NLBlockND(cin, cint, mode='dot', dimension=dim), seed=1).eval()
    x = torch.from_numpy(G[tag + "_x"])
    close(m(x), G[tag + "_y"])
    m.train()
    close(m(x), G[tag + "_y_train"])


def test_nlblock_dot_mode_without_the_pairwise_matrix_equals_the_literal_product():
    """mode='dot' with more positions than embedding channels takes theta^T ((phi g) / P); the literal (theta^T phi / P) g of
    reference backbone.py:673-690 gives the same values and gradients (float64: rounding only)."""
    from multimodal_gar_amd.model.backbone import NLBlockND
    from multimodal_gar_amd.nn_utils import conv1x1
    torch.manual_seed(0)
    m = fill_deterministic(NLBlockND(24, 3, mode='dot', dimension=3, bn_layer=False), seed=5).double()
    with torch.no_grad():
        m.W_z.weight.normal_()
    x = torch.randn(2, 24, 6, 6, 6, dtype=torch.float64, requires_grad=True)
    y = m(x)
    xr = x.detach().clone().requires_grad_(True)
    n, ci = 2, 3
    g_x = conv1x1(m.g, xr).view(n, ci, -1).permute(0, 2, 1)
    f = torch.matmul(conv1x1(m.theta, xr).view(n, ci, -1).permute(0, 2, 1), conv1x1(m.phi, xr).view(n, ci, -1))
    lit = torch.matmul(f / f.size(-1), g_x).permute(0, 2, 1).contiguous().view(n, ci, 6, 6, 6)
    want = conv1x1(m.W_z, lit) + xr
    assert f.shape == (2, 216, 216) and torch.allclose(y, want, rtol=1e-12, atol=1e-12)
    cot = torch.randn_like(y)
    got_g = torch.autograd.grad((y * cot).sum(), [x] + list(m.parameters()))
    want_g = torch.autograd.grad((want * cot).sum(), [xr] + list(m.parameters()))
    for a, b in zip(got_g, want_g):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)


def test_unit3d_same_padding_matches_reference():
    from multimodal_gar_amd.model.backbone import Unit3D
    u = fill_deterministic(Unit3D(3, 8, kernel_shape=[7, 7, 7], stride=(2, 2, 2)), seed=3).eval()
    with torch.no_grad():
        close(u(torch.from_numpy(G["unit3d_x"])), G["unit3d_y"])


def test_i3d_mixed4f_matches_reference():
    from multimodal_gar_amd.model.backbone import InceptionI3d
    m = InceptionI3d(final_endpoint='Mixed_4f'); m.build()
    assert sum(p.numel() for p in m.parameters()) == 7528256
    fill_deterministic(m, seed=2).eval()
    with torch.no_grad():
        y = m.extract_features(torch.from_numpy(G["i3d_x"]))
    assert y.shape == (1, 832, 4, 2, 3)
    close(y, G["i3d_y"])


def test_state_dict_keys_match_reference_names():
    """Checkpoint compatibility: spot-check the parameter names the reference's
    checkpoints carry."""
    from multimodal_gar_amd.model.backbone import InceptionI3d
    m = InceptionI3d(final_endpoint='Mixed_4f'); m.build()
    keys = set(m.state_dict())
    for k in ("Conv3d_1a_7x7.conv3d.weight", "Mixed_4f.b0.conv3d.weight", "Mixed_3b.b1b.bn.running_var",
              "Mixed_4b.b3b.bn.weight"):
        assert k in keys


# ------------------------------------------------------------------ GPU: blocks on HIP kernels
def _gar_cfg():
    from multimodal_gar_amd.pcdet.config import EasyDict
    return EasyDict(MODALITY="Multi", FUSION="Attention_mat", SIGMA=10, FEAT_NORM=True, EUCLIDEAN=True,
                    ind_action_concat=True, sg_feat_org=False, FEATURE_DIM=1024, HIDDEN_DIM=512, sim="cosine")


@pytest.mark.gpu
def test_dafm_layer_matches_reference():
    from multimodal_gar_amd.model.gat_model import FusionAttention_mat
    fa = fill_deterministic(FusionAttention_mat(input_dim=64, out_dim=64, sigma=10), seed=4).cuda().eval()
    R, L, De = (torch.from_numpy(G[k]).cuda() for k in ("dafm_R", "dafm_L", "dafm_De"))
    with torch.no_grad():
        Rp, Lp = fa(R, L, torch.zeros_like(De), De)
    close(Rp, G["dafm_Rp"]); close(Lp, G["dafm_Lp"])


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["per_scene", "batched"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_gar_fusion_net3_matches_reference(route, mode):
    from multimodal_gar_amd.model.gat_model import GAR_Fusion_Net3
    cfg = _gar_cfg()
    if route == "per_scene":
        cfg.DISABLE_BATCHED = True
    net = fill_deterministic(GAR_Fusion_Net3(cfg), seed=5).cuda()
    assert sum(p.numel() for p in net.parameters()) == 11063872
    rgb, lid, bb, b3 = (torch.from_numpy(G[k]).cuda() for k in ("gar_rgb", "gar_lidar", "gar_bboxes", "gar_bboxes3d"))
    pid = torch.from_numpy(G["gar_pid"]).cuda()
    if mode == "eval":
        net.eval()
    else:
        net.train()
        for mod in net.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    with torch.no_grad():
        res = net(rgb, lid, bb, b3, None, pid)
    assert len(res) == 16
    for i, r in enumerate(res):
        close(r, G["gar_%s_%02d" % (mode, i)], rtol=2e-4)
    if mode == "train":
        close(net.bn_rgb.running_mean, G["gar_train_bn_rgb_mean"])
        close(net.bn_rgb.running_var, G["gar_train_bn_rgb_var"])


@pytest.mark.gpu
def test_gar_fusion_net3_routes_agree_in_gradients():
    """The batched route must be the same function as the reference-shaped per-scene loop,
    including its gradients."""
    from multimodal_gar_amd.model.gat_model import GAR_Fusion_Net3
    rgb, lid, bb, b3 = (torch.from_numpy(G[k]).cuda() for k in ("gar_rgb", "gar_lidar", "gar_bboxes", "gar_bboxes3d"))
    pid = torch.from_numpy(G["gar_pid"]).cuda()
    grads = []
    for disable in (True, False):
        cfg = _gar_cfg(); cfg.DISABLE_BATCHED = disable
        net = fill_deterministic(GAR_Fusion_Net3(cfg), seed=5).cuda().train()
        for mod in net.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        r = rgb.clone().requires_grad_(True); l = lid.clone().requires_grad_(True)
        res = net(r, l, bb, b3, None, pid)
        loss = sum((o * o).sum() for o in res)
        loss.backward()
        grads.append((r.grad, l.grad, net.AttFusModule1.WQ_r.grad, net.D_embed[0].weight.grad, net.card_net[0].weight.grad))
    for a, b in zip(*grads):
        close(a, b.cpu().numpy(), rtol=5e-4, atol=1e-6)


@pytest.mark.gpu
def test_i3d_on_device_matches_reference_golden():
    """Device path of I3D: fused BN3d + ReLU (bn_act.hip) and fused same-padding max pool
    (maxpool3d.hip) against the reference's own output."""
    from multimodal_gar_amd.model.backbone import InceptionI3d
    m = InceptionI3d(final_endpoint='Mixed_4f'); m.build()
    fill_deterministic(m, seed=2).eval()
    m = m.cuda()
    with torch.no_grad():
        y = m.extract_features(torch.from_numpy(G["i3d_x"]).cuda())
    close(y, G["i3d_y"], rtol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k,s", [((2, 5, 15, 33, 47), (3, 3, 3), (2, 2, 2)), ((1, 7, 8, 20, 31), (1, 3, 3), (1, 2, 2)),
                                       ((2, 3, 4, 9, 10), (3, 3, 3), (1, 1, 1)), ((1, 2, 5, 6, 7), (2, 2, 2), (2, 2, 2)),
                                       # the vectorised kernel: W % 4 == 0 (stride 1) / W % 8 == 0 (stride 2)
                                       ((2, 3, 4, 9, 12), (3, 3, 3), (1, 1, 1)), ((1, 5, 3, 7, 160), (3, 3, 3), (1, 1, 1)),
                                       ((1, 4, 8, 22, 48), (1, 3, 3), (1, 2, 2)), ((2, 3, 7, 13, 16), (3, 3, 3), (2, 2, 2)),
                                       ((1, 2, 1, 1, 4), (3, 3, 3), (1, 1, 1)), ((1, 2, 2, 3, 8), (1, 3, 3), (1, 2, 2))])
def test_maxpool3d_same_padding_kernel(shape, k, s):
    from multimodal_gar_amd.model.backbone import MaxPool3dSamePadding
    torch.manual_seed(0)
    x = torch.randn(shape) - 0.5            # mostly negative: the zero padding must win where it is touched
    pool = MaxPool3dSamePadding(kernel_size=list(k), stride=s, padding=0)
    want = pool(x)                          # CPU: F.pad + MaxPool3d, the reference's op chain
    got = pool(x.cuda())
    assert got.shape == want.shape
    assert torch.equal(got.cpu(), want)


@pytest.mark.gpu
def test_i3d_batched_with_per_sample_bn_equals_one_pass_per_clip():
    """Several clips through I3D in one pass with per-SAMPLE BatchNorm statistics (bn_act.hip, grouped) must give
    what one train-mode pass per clip gives: features, running statistics (momentum updates in clip order) and the
    batch counter."""
    import copy
    from multimodal_gar_amd.model.backbone import InceptionI3d
    m = InceptionI3d(final_endpoint='Mixed_4f'); m.build()
    fill_deterministic(m, seed=5)
    a = m.cuda().train()
    b = copy.deepcopy(a)
    torch.manual_seed(1)
    x = torch.randn(3, 3, 7, 64, 96, device="cuda") * 2 + 0.3
    with torch.no_grad():
        want = torch.cat([a.extract_features(x[i:i + 1]) for i in range(3)])
        b.set_per_sample_stats(True)
        got = b.extract_features(x)
    close(got, want.cpu().numpy(), rtol=2e-4)
    bns_a = [mod for mod in a.modules() if isinstance(mod, torch.nn.BatchNorm3d)]
    bns_b = [mod for mod in b.modules() if isinstance(mod, torch.nn.BatchNorm3d)]
    assert len(bns_a) > 30
    for p, q in zip(bns_a, bns_b):
        close(q.running_mean, p.running_mean.cpu().numpy(), rtol=2e-4); close(q.running_var, p.running_var.cpu().numpy(), rtol=2e-4)
        assert int(q.num_batches_tracked) == int(p.num_batches_tracked) == 3
