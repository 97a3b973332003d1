"""Channels-last (NDHWC) forward kernels of the frozen I3D (csrc/channels_last.hpp) against the NCDHW kernels / torch on the
same values, and the whole I3D trunk in both layouts."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bn(c):
    bn = torch.nn.BatchNorm3d(c, eps=1e-3, momentum=0.01).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    return bn


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("shape", [(3, 24, 4, 9, 14), (2, 208, 2, 12, 20), (2, 832, 2, 6, 10), (1, 16, 8, 45, 80)])
def test_batchnorm_relu_channels_last_equals_ncdhw(shape, per_sample, dtype):
    from multimodal_gar_amd import bn_ops
    torch.manual_seed(1)
    c = shape[1]
    x = (torch.randn(*shape, device="cuda") * 0.7 + 30.0).to(dtype)             # a large common mean: cancellation-safe partials
    xcl = x.contiguous(memory_format=torch.channels_last_3d)
    a, b = _bn(c), _bn(c)
    b.load_state_dict(a.state_dict())
    fn = bn_ops.bn_act_per_sample if per_sample else bn_ops.bn_act
    with torch.no_grad():
        want = fn(x, a, True)
        wide = torch.zeros((shape[0], c + 8) + shape[2:], device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last_3d)
        got = fn(xcl, b, True, out=wide[:, 4:4 + c])
    assert got.data_ptr() == wide[:, 4:4 + c].data_ptr() and got.shape == want.shape
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    assert (got.float() - want.float()).abs().max().item() <= tol * (want.float().abs().max().item() + 1e-6)
    assert (wide[:, :4] == 0).all() and (wide[:, 4 + c:] == 0).all()
    assert torch.allclose(a.running_mean, b.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(a.running_var, b.running_var, rtol=1e-4, atol=1e-7)
    assert a.num_batches_tracked.item() == b.num_batches_tracked.item()
    # NCDHW in, channels-last out (the stem's BatchNorm)
    d = _bn(c)
    d.load_state_dict(a.state_dict())
    e = _bn(c)
    e.load_state_dict(a.state_dict())
    with torch.no_grad():
        t = fn(x, d, True, to_channels_last=True)
        ref = fn(x, e, True)
    assert t.is_contiguous(memory_format=torch.channels_last_3d) and (shape[1] == 1 or not t.is_contiguous())
    assert (t.float() - ref.float()).abs().max().item() <= tol * (ref.float().abs().max().item() + 1e-6)


@pytest.mark.parametrize("k,s", [((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (2, 2, 2))])
@pytest.mark.parametrize("shape", [(2, 64, 8, 23, 40), (1, 192, 5, 45, 37)])
def test_maxpool3d_same_channels_last_equals_ncdhw(shape, k, s):
    from multimodal_gar_amd.model.backbone import MaxPool3dSamePadding
    torch.manual_seed(2)
    x = torch.randn(*shape, device="cuda") - 0.5            # mostly negative borders: the zero padding must win there
    pool = MaxPool3dSamePadding(kernel_size=list(k), stride=s, padding=0)
    with torch.no_grad():
        want = pool(x)
        got = pool(x.contiguous(memory_format=torch.channels_last_3d))
    assert got.is_contiguous(memory_format=torch.channels_last_3d) and torch.equal(got, want)


def test_i3d_trunk_channels_last_equals_ncdhw():
    """InceptionI3d.extract_features, several clips per pass with per-clip statistics: NDHWC between the stem and the output
    == NCDHW, to fp32 rounding (convolution algorithms differ between the layouts)."""
    import copy
    from multimodal_gar_amd.model.backbone import InceptionI3d
    torch.manual_seed(3)
    net = InceptionI3d(final_endpoint='Mixed_4f')
    net.build()
    net = net.cuda().train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.8, 1.2)
                m.bias.normal_(0, 0.1)
    x = torch.randn(3, 3, 8, 64, 96, device="cuda")
    a, b = copy.deepcopy(net), copy.deepcopy(net)
    a.set_per_sample_stats(True)
    b.set_per_sample_stats(True)
    b.set_channels_last(True)
    with torch.no_grad():
        ya = a.extract_features(x)
        yb = b.extract_features(x)
    assert yb.shape == ya.shape and yb.is_contiguous(memory_format=torch.channels_last_3d)
    rel = ((yb - ya).pow(2).mean().sqrt() / ya.pow(2).mean().sqrt()).item()
    assert rel <= 2e-4, rel                                   # train-mode BatchNorm chains amplify conv rounding differences
    for (na, pa), (_, pb) in zip(a.named_buffers(), b.named_buffers()):
        if pa.is_floating_point():
            assert torch.allclose(pa, pb, rtol=2e-3, atol=1e-5), na
