"""csrc/conv3d_wino.hip (the 3x3x3 / stride-1 / "same" units of I3D: Winograd F(2,3) along W on the fp32 MFMA, NCDHW in and
out, padding in the kernel) against an fp64 convolution of the same fp32 operands, with the library's direct fp32 convolution
measured beside it: the kernel's error must stay within a small factor of the direct kernel's and within 5e-6 of the output
scale.  Shapes cover the three tile shapes (rows of 64 / 32 / 16 outputs), ragged H, partial channel groups (C_out not a
multiple of 64 or 32), one-plane volumes and the real I3D channel plans at reduced extent."""
import pytest
import torch
import torch.nn.functional as F

from conftest import record_error

pytestmark = pytest.mark.gpu

CASES = [
    # n, cin, cout, d, h, w
    (1, 16, 32, 3, 9, 20),        # 16-wide tiles, ragged H and W tiles
    (2, 24, 64, 4, 45, 80),       # Mixed_4c Branch_2 at its real extent
    (1, 6, 208, 2, 23, 160),      # 32-wide tiles; C_out = 3 full groups + a 16-channel tail
    (1, 64, 192, 2, 12, 320),     # Conv3d_2c_3x3's channel plan, 64-wide tiles
    (1, 2, 48, 1, 5, 6),          # one plane, one channel pair, tile larger than the image
    (3, 32, 96, 5, 17, 34),       # W = 34: a ragged last pair column
    (1, 160, 320, 2, 10, 16),     # Mixed_4f Branch_1's channel plan
]


def _conv(x, w):
    from multimodal_gar_amd import _lib as L
    n, cin, d, h, wd = x.shape
    cout = w.shape[0]
    y = torch.empty((n, cout, d, h, wd), dtype=torch.float32, device=x.device)
    wp = torch.empty((L.raw("mgar_conv3d_k3_workspace_floats", cin, cout),), dtype=torch.float32, device=x.device)
    L.call("mgar_conv3d_k3_fwd", L.fptr(x), n, cin, d, h, wd, L.fptr(w), cout, L.fptr(wp), L.fptr(y), L.stream_of(x))
    return y


@pytest.mark.parametrize("n,cin,cout,d,h,w", CASES)
def test_conv3d_k3_against_float64(n, cin, cout, d, h, w):
    g = torch.Generator().manual_seed(cin * 1000 + cout + w)
    # activations like the trunk's: post-ReLU (non-negative, many zeros), weights centred
    x = torch.relu(torch.randn(n, cin, d, h, w, generator=g) + 0.3)
    wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
    truth = F.conv3d(x.double(), wt.double(), None, 1, 1)
    xg, wg = x.cuda(), wt.cuda()
    got = _conv(xg, wg).cpu().double()
    lib = F.conv3d(xg, wg, None, 1, 1).cpu().double()
    scale = truth.abs().max().item()
    e_k, e_l = (got - truth).abs().max().item(), (lib - truth).abs().max().item()
    record_error("conv3d_k3 wino vs fp64", e_k, scale, 5e-6)
    record_error("conv3d_k3 library vs fp64", e_l, scale, 5e-6)
    assert e_k <= 5e-6 * scale, (e_k, e_l, scale)
    assert e_k <= max(4.0 * e_l, 2e-6 * scale), "Winograd error %g vs direct %g (scale %g)" % (e_k, e_l, scale)
    # rms error as well (the maximum is one element)
    r_k = (got - truth).pow(2).mean().sqrt().item()
    r_l = (lib - truth).pow(2).mean().sqrt().item()
    record_error("conv3d_k3 wino rms vs fp64", r_k, scale, 5e-6)
    record_error("conv3d_k3 library rms vs fp64", r_l, scale, 5e-6)
    assert r_k <= max(4.0 * r_l, 5e-7 * scale)


def test_conv3d_k3_rejects_odd_shapes():
    from multimodal_gar_amd import _lib as L
    x = torch.zeros(1, 3, 2, 4, 6, device="cuda")
    w = torch.zeros(8, 3, 3, 3, 3, device="cuda")
    with pytest.raises(Exception):
        _conv(x, w)
    x = torch.zeros(1, 4, 2, 4, 7, device="cuda")
    w = torch.zeros(8, 4, 3, 3, 3, device="cuda")
    with pytest.raises(Exception):
        _conv(x, w)
    assert L.raw("mgar_conv3d_k3_workspace_floats", 4, 8) == 64 * 4 * 36


def test_unit3d_takes_the_kernel_and_matches_the_library():
    """Unit3D's 3x3x3 units go through the kernel on the device when frozen (no autograd), and give the library's result up to
    fp32 summation order; with gradients required they stay on the library path."""
    from multimodal_gar_amd.model.backbone import Unit3D
    torch.manual_seed(5)
    u = Unit3D(32, 96, [3, 3, 3], name="t").cuda().train()
    x = torch.relu(torch.randn(2, 32, 4, 21, 40, device="cuda"))
    with torch.no_grad():
        z = u._conv(x)
        assert u._k3_conv(x) is not None
        u.wino_kernel = False
        assert u._k3_conv(x) is None
        z_lib = u._conv(x)
        u.wino_kernel = True
    scale = z_lib.abs().max().item()
    err = (z - z_lib).abs().max().item()
    record_error("Unit3D k3 kernel vs library", err, scale, 5e-6)
    assert err <= 5e-6 * scale
    xr = x.clone().requires_grad_(True)
    assert u._k3_conv(xr) is None          # a trained trunk keeps the library convolution (autograd)


@pytest.mark.parametrize("shape", [(1, 5, 24, 640), (2, 15, 64, 96), (1, 4, 30, 132), (1, 2, 10, 8)])
def test_stem_minimal_filtering_against_float64_and_the_direct_kernel(shape):
    """csrc/stem_conv.hip, fp32, W % 4 == 0: the parity-split / F(2,3) variant against the fp64 convolution, with the direct
    kernel (switch off) measured beside it -- same error class; wide rows (5 column tiles), ragged tiles, tiny volumes."""
    from multimodal_gar_amd import _lib as L
    from multimodal_gar_amd.model.backbone import Unit3D
    n, t, h, w = shape
    torch.manual_seed(11)
    u = Unit3D(3, 64, [7, 7, 7], stride=(2, 2, 2), padding=(3, 3, 3), use_batch_norm=False, activation_fn=None).cuda()
    x = torch.randn(n, 3, t, h, w, device="cuda")
    pads = []
    for size in (w, h, t):
        total = 5 if size % 2 == 0 else 6
        pads += [total // 2, total - total // 2]
    with torch.no_grad():
        want = F.conv3d(F.pad(x.double().cpu(), pads), u.conv3d.weight.double().cpu(), stride=2)
        got = u(x).double().cpu()
        L.call("mgar_stem_conv3d_set_minimal_filtering", 0)
        try:
            direct = u(x).double().cpu()
        finally:
            L.call("mgar_stem_conv3d_set_minimal_filtering", 1)
    scale = want.abs().max().item()
    e_k, e_d = (got - want).abs().max().item(), (direct - want).abs().max().item()
    record_error("stem minimal filtering vs fp64", e_k, scale, 5e-6)
    record_error("stem direct kernel vs fp64", e_d, scale, 5e-6)
    assert got.shape == want.shape
    assert e_k <= 5e-6 * scale, (e_k, e_d, scale)
    assert e_k <= max(4.0 * e_d, 2e-6 * scale)
    assert (got - direct).abs().max().item() > 0 or n * t * h * w < 1000     # (the two kernels really are different code paths)


def test_i3d_trunk_with_own_convolutions_matches_the_library_route():
    """InceptionI3d.extract_features (to Mixed_4f, frozen, train-mode BatchNorm with per-clip statistics) with the round-3 routes --
    minimal-filtering stem, F(2,3) 3x3x3 units, 1x1x1 units as batched GEMMs -- against the same module on the library's
    convolutions: 17 convolution + BatchNorm levels deep, the two outputs agree to 3.5e-5 of the feature scale (tolerance 1e-4)."""
    from multimodal_gar_amd import _lib as L
    from multimodal_gar_amd.model.backbone import InceptionI3d, Unit3D
    torch.manual_seed(21)
    net = InceptionI3d(final_endpoint="Mixed_4f")
    net.build()
    net = net.cuda().train()
    net.set_per_sample_stats(True)
    x = torch.randn(2, 3, 9, 96, 160, device="cuda")
    units = [m for m in net.modules() if isinstance(m, Unit3D)]
    with torch.no_grad():
        own = net.extract_features(x)
        for u in units:
            u.wino_kernel = False
            u.gemm_1x1 = False
        L.call("mgar_stem_conv3d_set_minimal_filtering", 0)
        try:
            lib = net.extract_features(x)
        finally:
            L.call("mgar_stem_conv3d_set_minimal_filtering", 1)
            for u in units:
                u.wino_kernel = True
                u.gemm_1x1 = True
    assert own.shape == lib.shape == (2, 832, 3, 6, 10)
    scale = lib.abs().max().item()
    err = (own - lib).abs().max().item()
    record_error("I3D to Mixed_4f: own convolutions vs library route", err, scale, 1e-4)
    assert err <= 1e-4 * scale, (err, scale)
