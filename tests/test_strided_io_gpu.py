"""Kernels that read / write a CHANNEL SLICE (or a transposed view) of a wider tensor in place, instead of going through a
torch.cat / .contiguous() copy (round 2: mgar_bn_act_fwd_into, mgar_three_interpolate_grad_*_strided,
mgar_bn_act_maxpool_bwd_strided).  Each must give, bit for bit, what the contiguous entry point gives on a copy."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("per_sample", [False, True])
def test_bn_act_writes_into_a_channel_slice(dtype, per_sample):
    from multimodal_gar_amd import bn_ops
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm3d(24, eps=1e-3, momentum=0.01).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    x = (torch.randn(3, 24, 4, 10, 12, device="cuda") * 2 + 1).to(dtype)
    fn = bn_ops.bn_act_per_sample if per_sample else bn_ops.bn_act
    with torch.no_grad():
        want = fn(x, bn, True)
        wide = torch.full((3, 64, 4, 10, 12), -7.0, device="cuda", dtype=dtype)
        dst = wide[:, 16:40]
        got = fn(x, bn, True, out=dst)
    assert got is dst and torch.equal(dst, want)
    assert (wide[:, :16] == -7).all() and (wide[:, 40:] == -7).all(), "wrote outside its channels"
    # a destination that is not such a slice is ignored (the caller copies)
    with torch.no_grad():
        other = fn(x, bn, True, out=wide[:, 16:40].transpose(3, 4))
    assert other.shape == x.shape and torch.equal(other, want)


def test_inception_module_without_cat_equals_cat():
    from multimodal_gar_amd.model.backbone import InceptionModule
    torch.manual_seed(1)
    mod = InceptionModule(32, [16, 24, 32, 8, 16, 16], "Mixed_test").cuda().train()
    x = torch.randn(2, 32, 4, 14, 18, device="cuda")
    with torch.no_grad():
        got = mod(x)                                             # device path: branches write into the concatenated tensor
    xr = x.clone().requires_grad_(True)                         # autograd path: torch.cat
    want = mod(xr)
    assert got.shape == want.shape == (2, 16 + 32 + 16 + 16, 4, 14, 18)
    assert (got - want.detach()).abs().max().item() <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("n,m,c", [(4096, 1024, 64), (1000, 37, 20), (40000, 5000, 8)])   # sorted path, odd sizes, LDS / atomic path
def test_three_interpolate_backward_reads_a_channel_slice_in_place(n, m, c):
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as U
    torch.manual_seed(2)
    b = 3
    known = torch.randn(b, c, m, device="cuda")
    idx = torch.randint(0, m, (b, n, 3), device="cuda", dtype=torch.int32)
    w = torch.rand(b, n, 3, device="cuda")
    w = w / w.sum(2, keepdim=True)
    skip = torch.randn(b, 8, n, device="cuda")
    res = []
    for sliced in (True, False):
        k = known.clone().requires_grad_(True)
        spread = U.three_interpolate(k, idx, w)
        merged = torch.cat([spread, skip], 1)                    # as PointnetFPModule does (pointnet2_modules.py:139-148)
        cot = torch.linspace(-1, 1, merged.numel(), device="cuda").view(merged.shape)
        if sliced:
            (merged * cot).sum().backward()                      # grad of `spread` = a channel slice of grad(merged)
        else:
            spread.backward(cot[:, :c].contiguous())
        res.append(k.grad)
    if n <= 36864:
        assert torch.equal(res[0], res[1])                       # inverted-index path: fixed summation order
    else:                                                        # LDS-atomic path: the order of the adds is not fixed
        assert (res[0] - res[1]).abs().max().item() <= 1e-5 * res[1].abs().max().item()


@pytest.mark.parametrize("layout", ["channel_slice", "transposed_rows"])
def test_bn_maxpool_backward_reads_a_strided_gradient_in_place(layout):
    from multimodal_gar_amd import bn_ops
    torch.manual_seed(3)
    b, c, m, ns = (2, 24, 500, 16) if layout == "channel_slice" else (1, 32, 3000, 16)
    bn = torch.nn.BatchNorm2d(c).cuda().train()
    x = torch.randn(b, c, m, ns, device="cuda")
    if layout == "channel_slice":
        wide = torch.randn(b, 70, m, device="cuda")
        g = wide[:, 10:10 + c]
    else:
        rows = torch.randn(m, 96, device="cuda")                 # (M, C_total) rows; the pooled gradient is a transposed slice
        g = rows.t()[32:32 + c].unsqueeze(0)
    assert not g.is_contiguous()
    res = []
    for grad in (g, g.contiguous()):
        xx = x.clone().requires_grad_(True)
        bn.zero_grad()
        y = bn_ops.bn_act_maxpool(xx, bn, True)
        y.backward(grad)
        res.append((xx.grad, bn.weight.grad.clone(), bn.bias.grad.clone()))
    for a, bb in zip(*res):
        assert torch.equal(a, bb)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_three_interpolate_concat_equals_interpolate_then_cat(dtype):
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as U
    torch.manual_seed(5)
    b, c, m, n, cs = 3, 40, 600, 2400, 9
    known = torch.randn(b, c, m, device="cuda").to(dtype)
    skip = torch.randn(b, cs, n, device="cuda").to(dtype)
    idx = torch.randint(0, m, (b, n, 3), device="cuda", dtype=torch.int32)
    w = torch.rand(b, n, 3, device="cuda")
    w = w / w.sum(2, keepdim=True)
    with torch.no_grad():
        got = U.three_interpolate_concat(known, idx, w, skip)
        want = torch.cat([U.three_interpolate(known, idx, w), skip], 1)
    assert got.dtype == dtype and torch.equal(got, want)
    if dtype == torch.float32:                                   # gradients (the bf16 path is forward-only)
        res = []
        for fused in (True, False):
            k, s = known.clone().requires_grad_(True), skip.clone().requires_grad_(True)
            y = U.three_interpolate_concat(k, idx, w, s) if fused else torch.cat([U.three_interpolate(k, idx, w), s], 1)
            (y * torch.linspace(-1, 1, y.numel(), device="cuda").view(y.shape)).sum().backward()
            res.append((k.grad, s.grad))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("shape", [(3, 24, 2, 23, 40), (1, 64, 4, 44, 80), (2, 8, 1, 1, 4)])
def test_small_channel_batchnorm_in_one_launch_equals_three_launches(dtype, per_sample, shape, monkeypatch):
    """bn_small_fused_kernel (statistics + apply in one launch, <= 16 384 elements per channel) against the partial / finalize /
    apply path: outputs, running statistics, writing into a channel slice; also with a large common mean."""
    from multimodal_gar_amd import bn_ops
    torch.manual_seed(7)
    c = shape[1]

    def make():
        bn = torch.nn.BatchNorm3d(c, eps=1e-3, momentum=0.01).cuda().train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_()
        return bn
    x = (torch.randn(*shape, device="cuda") * 0.5 + 40.0).to(dtype)
    fn = bn_ops.bn_act_per_sample if per_sample else bn_ops.bn_act
    small, plain = make(), make()
    plain.load_state_dict(small.state_dict())
    with torch.no_grad():
        wide = torch.zeros((shape[0], c + 8) + shape[2:], device="cuda", dtype=dtype)
        got = fn(x, small, True, out=wide[:, 4:4 + c])
        monkeypatch.setattr(bn_ops, "SMALL_CHANNEL_MAX", 0)
        want = fn(x, plain, True)
    assert got.data_ptr() == wide[:, 4:4 + c].data_ptr()
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5                # bf16: one ulp of the stored result where rounding flips
    assert (got.float() - want.float()).abs().max().item() <= tol * (want.float().abs().max().item() + 1e-6)
    assert (wide[:, :4] == 0).all() and (wide[:, 4 + c:] == 0).all()
    assert torch.allclose(small.running_mean, plain.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(small.running_var, plain.running_var, rtol=1e-4, atol=1e-7)
    assert small.num_batches_tracked.item() == plain.num_batches_tracked.item() == (shape[0] if per_sample else 1)
