"""BatchNorm statistics partials left behind by the producer kernels (round 2): mgar_pointwise_conv_fwd_stats and
mgar_query_group_proj_stack_fwd_stats write, per channel and 128-column tile of their output, the tile's mean and sum of
squared deviations; mgar_bn_stats_from_partials merges them (the finalize of mgar_bn_train_stats) -- the BatchNorm that
follows needs no pass over the tensor.  Oracle: float64 mean / biased variance of the tensor the kernel wrote."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(mean, invstd, y, eps, what):
    y64 = y.double()
    m = y64.mean((0, 2))
    v = y64.var((0, 2), unbiased=False)
    assert (mean.double() - m).abs().max().item() <= 2e-6 * (m.abs().max().item() + v.sqrt().max().item()), what + " mean"
    want = 1.0 / (v + eps).sqrt()
    assert ((invstd.double() - want).abs() / want).max().item() <= 2e-6, what + " invstd"


@pytest.mark.parametrize("B,cin,cout,P,offset", [(1, 32, 32, 128 * 700, 0.0), (3, 16, 24, 128 * 200, 0.0), (2, 32, 32, 128 * 320, 300.0)])
def test_pointwise_conv_statistics_epilogue(B, cin, cout, P, offset):
    from multimodal_gar_amd import bn_ops
    torch.manual_seed(0)
    bn_in = torch.nn.BatchNorm2d(cin).cuda().train()
    bn_out = torch.nn.BatchNorm2d(cout).cuda().train()
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    x = (torch.randn(B, cin, P // 16, 16, device="cuda") * 1.5 + 0.3)
    if offset:                                                   # a large common mean: the partials must stay cancellation-safe
        with torch.no_grad():
            bn_in.bias.fill_(offset)
            bn_in.weight.fill_(1.0)
    with torch.no_grad():
        y, stats = bn_ops.bn_act_conv(x, bn_in, False if offset else True, conv, want_out_stats=True)
        assert stats is not None and stats.shape == (cout, B * P // 128, 2)
        plain = bn_ops.bn_act_conv(x, bn_in, False if offset else True, conv)
        assert torch.equal(y, plain)                             # the epilogue does not touch the product
        y3 = y.flatten(2)
        ref = torch.nn.BatchNorm2d(cout).cuda().train()
        m_ref, is_ref = bn_ops._train_stats(y3, ref)             # the separate statistics pass
        m_fused, is_fused = bn_ops._train_stats(y3, bn_out, stats)
    _check(m_fused, is_fused, y3, bn_out.eps, "fused")
    _check(m_ref, is_ref, y3, ref.eps, "separate")
    assert torch.allclose(bn_out.running_mean, ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn_out.running_var, ref.running_var, rtol=1e-5, atol=1e-6)
    assert bn_out.num_batches_tracked.item() == 1


def test_stack_grouping_statistics_epilogue_and_module_equivalence():
    """StackSAModuleMSG with the partials handed from kernel to kernel == the same module with every BatchNorm computing its
    own statistics (outputs, gradients, running statistics), to fp32 rounding."""
    import copy
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from param_fill import fill_deterministic
    from multimodal_gar_amd import bn_ops
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as MS
    torch.manual_seed(4)
    n = 3000
    sx = (torch.rand(2 * n, 3) * torch.tensor([8.0, 8.0, 2.0])).cuda()
    cnt = torch.tensor([n, n], dtype=torch.int32, device="cuda")
    new_xyz = torch.cat([sx[:2200] + 0.03, sx[n:n + 2152] - 0.02]).contiguous()        # 4352 queries x 16 = 128 * 544 columns
    ncnt = torch.tensor([2200, 2152], dtype=torch.int32, device="cuda")
    feats = torch.randn(2 * n, 40, device="cuda")
    mod = fill_deterministic(MS.StackSAModuleMSG(radii=[0.6, 1.2], nsamples=[16, 16], mlps=[[40, 32, 32], [40, 24, 32]]), seed=8).cuda().train()
    seen = []
    real = bn_ops._train_stats

    def spy(x3, bn, partial=None):
        seen.append(partial is not None)
        return real(x3, bn, partial)

    def run(fused):
        m = copy.deepcopy(mod)
        f = feats.clone().requires_grad_(True)
        old = bn_ops.stats_partial_buffer
        if not fused:
            bn_ops.stats_partial_buffer = lambda *a, **k: None
        bn_ops._train_stats = spy
        try:
            _, y = m(sx, cnt, new_xyz, ncnt, f)
        finally:
            bn_ops.stats_partial_buffer = old
            bn_ops._train_stats = real
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view(y.shape)).sum().backward()
        return y.detach(), f.grad, [p.grad for p in m.parameters()], [b for b in m.buffers() if b.is_floating_point()]

    seen.clear()
    a = run(True)
    assert seen == [True, True, True, True], seen               # both BatchNorms of both scales took the producer's partials
    seen.clear()
    b = run(False)
    assert seen == [False] * 4

    def close(p, q, what, rtol=2e-5):
        err, scale = (p - q).abs().max().item(), q.abs().max().item()
        assert err <= rtol * scale + 1e-7, (what, err, scale)
    close(a[0], b[0], "output")
    close(a[1], b[1], "d features", 2e-4)
    for i, (p, q) in enumerate(zip(a[2], b[2])):
        close(p, q, "param %d" % i, 2e-4)
    for i, (p, q) in enumerate(zip(a[3], b[3])):
        close(p, q, "buffer %d" % i)


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("B,cin,cout,P", [(1, 32, 32, 128 * 600), (3, 16, 24, 4 * 7000), (2, 64, 64, 4 * 9001), (2, 7, 5, 4 * 10000)])
def test_fused_bn_relu_conv_backward_with_reduction_from_the_dw_kernel(B, cin, cout, P, relu, monkeypatch):
    """_BnActConv.backward with dW and the BatchNorm-backward reduction from one pass (csrc/pointwise_dw.hip pair mode) against
    the three-pass form (dW; reduce; apply) and against float64 autograd of the same expression."""
    from multimodal_gar_amd import bn_ops
    torch.manual_seed(11)
    bn = torch.nn.BatchNorm2d(cin).cuda().train()
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    x = (torch.randn(B, cin, P // 4, 4, device="cuda") * 1.3 + 0.4)
    cot = torch.randn(B, cout, P // 4, 4, device="cuda")
    res = []
    for fused in (True, False):
        monkeypatch.setattr(bn_ops, "FUSED_DW_BN_REDUCE", fused)
        xx = x.clone().requires_grad_(True)
        bn.zero_grad(); conv.zero_grad()
        y = bn_ops.bn_act_conv(xx, bn, relu, conv)
        assert y is not None
        y.backward(cot)
        res.append((xx.grad.clone(), conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
    # float64 reference
    x64 = x.double().requires_grad_(True)
    g64, b64, w64 = (bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True),
                     conv.weight.detach().double().requires_grad_(True))
    z = torch.nn.functional.batch_norm(x64, None, None, g64, b64, True, 0.0, bn.eps)
    z = torch.relu(z) if relu else z
    torch.nn.functional.conv2d(z, w64).backward(cot.double())
    want = (x64.grad, w64.grad, g64.grad, b64.grad)
    for name, got, ref, w in zip(("dx", "dW", "dgamma", "dbeta"), res[0], res[1], want):
        scale = w.abs().max().item() + 1e-12
        assert (got.double() - w).abs().max().item() <= 2e-4 * scale, (name, "fused vs float64")
        assert (got - ref).abs().max().item() <= 2e-4 * scale, (name, "fused vs three-pass")


@pytest.mark.parametrize("B,cin,cout,P", [(2, 4, 32, 128 * 300), (1, 4, 16, 128 * 600), (3, 7, 24, 4 * 20001), (1, 32, 32, 128 * 512)])
def test_first_layer_conv_on_the_streaming_kernel(B, cin, cout, P):
    """The first conv of a shared MLP (no BatchNorm in front) on csrc/pointwise_fwd.hip with the identity activation:
    product, statistics partials for the BatchNorm behind it, both gradients -- against torch's convolution."""
    from multimodal_gar_amd import bn_ops
    torch.manual_seed(1)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    x = (torch.randn(B, cin, P // 4, 4, device="cuda") * 2 + 0.5).requires_grad_(True)
    out = bn_ops.plain_conv(x, conv, want_out_stats=True)
    assert out is not None
    y, stats = out
    xr = x.detach().clone().requires_grad_(True)
    ref = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    ref.load_state_dict(conv.state_dict())
    want = ref(xr)
    scale = want.abs().max().item()
    assert y.shape == want.shape and (y - want).abs().max().item() <= 2e-6 * scale
    if P % 128 == 0:
        assert stats is not None and stats.shape == (cout, B * P // 128, 2)
        bn = torch.nn.BatchNorm2d(cout).cuda().train()
        with torch.no_grad():
            m, istd = bn_ops._train_stats(y.detach().flatten(2), bn, stats)
        _check(m, istd, y.detach().flatten(2), bn.eps, "first layer")
    else:
        assert stats is None
    cot = torch.linspace(-1, 1, y.numel(), device="cuda").view(y.shape)
    (y * cot).sum().backward()
    (want * cot).sum().backward()
    assert (x.grad - xr.grad).abs().max().item() <= 1e-5 * xr.grad.abs().max().item()
    assert (conv.weight.grad - ref.weight.grad).abs().max().item() <= 2e-5 * ref.weight.grad.abs().max().item()
    # outside the kernel's shapes the caller keeps the library GEMM
    assert bn_ops.plain_conv(torch.randn(1, 4, 100, 4, device="cuda"), conv if cin == 4 else torch.nn.Conv2d(4, 8, 1, bias=False).cuda()) is None
    assert bn_ops.plain_conv(torch.randn(1, 40, 1 << 16, 4, device="cuda"), torch.nn.Conv2d(40, 16, 1, bias=False).cuda()) is None


def test_shared_mlp_with_first_layer_kernel_equals_unfused_module():
    from multimodal_gar_amd.nn_utils import PointwiseSequential
    import copy
    torch.manual_seed(2)
    mlp = PointwiseSequential(torch.nn.Conv2d(4, 16, 1, bias=False), torch.nn.BatchNorm2d(16), torch.nn.ReLU(),
                              torch.nn.Conv2d(16, 32, 1, bias=False), torch.nn.BatchNorm2d(32), torch.nn.ReLU()).cuda().train()
    plain = copy.deepcopy(mlp)
    plain.fuse_bn_conv = False                                   # every BatchNorm computes its own statistics, convs on the library
    x = torch.randn(2, 4, 4096, 16, device="cuda")
    res = []
    for m in (mlp, plain):
        y = m.forward_maxpool(x.clone())
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view(y.shape)).sum().backward()
        res.append((y.detach(), [p.grad.clone() for p in m.parameters()], [b.clone() for b in m.buffers()]))
    assert (res[0][0] - res[1][0]).abs().max().item() <= 1e-4 * res[1][0].abs().max().item()
    for a, b in zip(res[0][1], res[1][1]):
        assert (a - b).abs().max().item() <= 5e-4 * (b.abs().max().item() + 1e-6)
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.allclose(a.float(), b.float(), rtol=1e-4, atol=1e-6)
