"""Row-major BatchNorm1d(train) [+ ReLU] of the sparse trunk's (N_active, C) features (csrc/channels_last.hpp, round 3:
bn_cl_* forward, bn_rows_bwd_* backward) against torch's BatchNorm1d in float64: output, running statistics, dx, dgamma,
dbeta; and SparseSequential takes that path on the device."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,c,relu", [(100003, 32, True), (4097, 64, True), (1000, 128, False), (7, 16, True), (250000, 16, True)])
def test_bn_act_rows_matches_float64(rows, c, relu):
    from multimodal_gar_amd import bn_ops
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * 2.0 + 0.7)
    bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5); bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
    ref = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    bn = bn.cuda().train(); ref.train()
    xg = x.cuda().requires_grad_(True)
    y = bn_ops.bn_act_rows(xg, bn, relu)
    assert y is not None and y.shape == (rows, c)
    xr = x.double().requires_grad_(True)
    pre = ref(xr)
    yr = torch.relu(pre) if relu else pre
    # elements whose pre-activation is within 1e-4 of zero may sit on either side of the ReLU in fp32: keep them out of the cotangent
    cot = torch.randn(rows, c, generator=g, dtype=torch.float64) * ((pre.detach().abs() > 1e-4) if relu else 1.0)
    (y.double() * cot.cuda()).sum().backward()
    (yr * cot).sum().backward()

    def close(a, b, tol, what):
        a, b = a.detach().double().cpu(), b.detach().double()
        err, scale = (a - b).abs().max().item(), b.abs().max().item() + 1e-12
        assert err <= tol * scale + 1e-7, "%s: err %g scale %g" % (what, err, scale)
    close(y, yr, 2e-5, "y")
    close(bn.running_mean, ref.running_mean, 1e-6, "running_mean")
    close(bn.running_var, ref.running_var, 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == 1
    close(xg.grad, xr.grad, 1e-4, "dx")
    close(bn.weight.grad, ref.weight.grad, 1e-4, "dgamma")
    close(bn.bias.grad, ref.bias.grad, 1e-4, "dbeta")


def test_sparse_sequential_uses_the_row_major_kernels():
    from multimodal_gar_amd import _lib as L
    from multimodal_gar_amd.pcdet.utils.spconv_utils import spconv
    seq = spconv.SparseSequential(nn.BatchNorm1d(32, eps=1e-3, momentum=0.01), nn.ReLU()).cuda().train()
    x = spconv.SparseConvTensor(torch.randn(5000, 32, device="cuda"), torch.zeros(5000, 4, dtype=torch.int32, device="cuda"), [4, 4, 4], 1)
    calls = []
    real = L.call
    L.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        y = seq(x)
    finally:
        L.call = real
    assert "mgar_bn_cl_train_stats" in calls and "mgar_bn_cl_act_fwd" in calls
    want = torch.relu(nn.functional.batch_norm(x.features, None, None, seq[0].weight, seq[0].bias, True, 0.0, 1e-3))
    assert torch.allclose(y.features, want, rtol=1e-4, atol=1e-5)
