"""Sparse 3-D convolution stack (the spconv trunk of Voxel R-CNN): voxel hash table, rulebook,
gather-GEMM forward / data gradient / weight gradient (csrc/sparse_conv.hip) and VoxelBackBone8x built on them.

Oracle (tests only): the DENSE conv3d of the densified tensor read back at the active sites
(oracle/cpu_backend.py::sparse_conv3d_dense) -- the definition of a sparse convolution, in plain torch on CPU, float64
for the op-level checks.  spconv itself is third-party, absent from the reference tree and not installable here: parity
with it is unpinned by construction."""
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


def sparse_sites(seed, batch, shape, density):
    rng = np.random.default_rng(seed)
    coords = []
    for b in range(batch):
        occ = rng.random(shape) < density
        z, y, x = np.nonzero(occ)
        perm = rng.permutation(len(z))                        # rows in arbitrary order, as a voxeliser produces them
        coords.append(np.stack([np.full(len(z), b), z[perm], y[perm], x[perm]], 1))
    return torch.from_numpy(np.concatenate(coords).astype(np.int32))


def close(a, b, rtol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= rtol * (b.abs().max().item() + 1e-12) + 1e-7, "%s: max err %g vs scale %g" % (what, err, b.abs().max().item())


def test_voxel_hash_lookup_matches_dense_table():
    from multimodal_gar_amd.sparse_ops import VoxelHash
    shape = (7, 33, 29)
    idx = sparse_sites(1, 3, shape, 0.2).cuda()
    h = VoxelHash(idx, shape)
    dense = -torch.ones((3,) + shape, dtype=torch.int32)
    c = idx.cpu().long()
    dense[c[:, 0], c[:, 1], c[:, 2], c[:, 3]] = torch.arange(idx.shape[0], dtype=torch.int32)
    rng = np.random.default_rng(2)
    q = np.stack([rng.integers(0, 3, 5000), rng.integers(-2, shape[0] + 2, 5000), rng.integers(-2, shape[1] + 2, 5000),
                  rng.integers(-2, shape[2] + 2, 5000)], 1).astype(np.int32)
    got = h.lookup(torch.from_numpy(q).cuda()).cpu()
    inside = (q[:, 1] >= 0) & (q[:, 1] < shape[0]) & (q[:, 2] >= 0) & (q[:, 2] < shape[1]) & (q[:, 3] >= 0) & (q[:, 3] < shape[2])
    want = np.full(5000, -1, np.int32)
    qi = q[inside]
    want[inside] = dense[qi[:, 0], qi[:, 1], qi[:, 2], qi[:, 3]].numpy()
    assert np.array_equal(got.numpy(), want)
    assert (want >= 0).sum() > 200


CASES = [  # (subm, kernel, stride, padding, cin, cout)
    (True, 3, 1, 1, 4, 16), (True, 3, 1, 1, 16, 16), (True, 3, 1, 1, 64, 64), (True, 3, 1, 1, 33, 70),
    (False, 3, 2, 1, 16, 32), (False, 3, 2, (0, 1, 1), 64, 64), (False, (3, 1, 1), (2, 1, 1), 0, 64, 128), (False, 2, 2, 0, 8, 24),
]


@pytest.mark.parametrize("subm,kernel,stride,padding,cin,cout", CASES)
def test_sparse_conv_forward_backward_vs_dense_oracle(subm, kernel, stride, padding, cin, cout):
    from multimodal_gar_amd import sparse_ops
    from oracle.cpu_backend import sparse_conv3d_dense
    shape, batch = [9, 20, 24], 2
    idx = sparse_sites(3, batch, tuple(shape), 0.12)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(idx.shape[0], cin, generator=g)
    kk = sparse_ops._triple(kernel)
    w = torch.randn(cout, *kk, cin, generator=g) / (cin * kk[0] * kk[1] * kk[2]) ** 0.5
    f64, w64 = feats.double().requires_grad_(True), w.double().requires_grad_(True)
    want, widx, wshape = sparse_conv3d_dense(f64, idx, shape, batch, w64, kernel, stride, padding, subm, {}, None)
    cot = torch.linspace(-1, 1, want.numel(), dtype=torch.float64).view(want.shape)
    (want * cot).sum().backward()
    fg, wg = feats.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    got, gidx, gshape = sparse_ops.sparse_conv3d(fg, idx.cuda(), shape, batch, wg, kernel, stride, padding, subm, {}, "k")
    assert list(gshape) == list(wshape) and torch.equal(gidx.cpu(), widx), "output sites (ascending b, z, y, x) differ"
    (got * cot.float().cuda()).sum().backward()
    close(got, want, 2e-5, "out")
    close(fg.grad, f64.grad, 2e-5, "d features")
    close(wg.grad, w64.grad, 2e-5, "d weight")
    assert got.shape[0] > 50 and want.abs().sum() > 0


def test_voxel_backbone8x_state_dict_and_parity_vs_oracle_backend():
    """VoxelBackBone8x (reference spconv_backbone.py:69-170): parameter names / shapes of the reference, and the device run
    (sparse kernels) against the oracle backend (dense conv3d), train mode, outputs + input / parameter gradients."""
    from multimodal_gar_amd.pcdet.config import EasyDict
    from multimodal_gar_amd.pcdet.models.backbones_3d import VoxelBackBone8x
    from oracle.cpu_backend import use_cpu_oracle
    grid = [48, 40, 40]                                   # x, y, z cells -> sparse shape (41, 40, 48): z 41 -> 21 -> 11 -> 5 -> 2
    net = fill_deterministic(VoxelBackBone8x(EasyDict(NAME="VoxelBackBone8x"), 4, grid), seed=3).train()
    sd = net.state_dict()
    assert tuple(sd["conv_input.0.weight"].shape) == (16, 3, 3, 3, 4)          # spconv 2.x layout (C_out, kz, ky, kx, C_in)
    assert tuple(sd["conv2.0.0.weight"].shape) == (32, 3, 3, 3, 16) and tuple(sd["conv_out.0.weight"].shape) == (128, 3, 1, 1, 64)
    assert "conv4.2.1.running_var" in sd and "conv_out.1.weight" in sd and len([k for k in sd if k.endswith(".weight")]) == 24
    idx = sparse_sites(7, 2, (40, 40, 48), 0.03)
    g = torch.Generator().manual_seed(8)
    feats = torch.randn(idx.shape[0], 4, generator=g)
    res = {}
    for dev in ("cuda", "cpu"):
        m = copy.deepcopy(net).to(dev)
        f = feats.to(dev).clone().requires_grad_(True)
        data = {"batch_size": 2, "voxel_features": f, "voxel_coords": idx.to(dev)}
        ctx = use_cpu_oracle() if dev == "cpu" else None
        if ctx:
            ctx.__enter__()
        try:
            out = m(data)
            loss = 0
            for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
                t = out["multi_scale_3d_features"][name].features
                loss = loss + (t * torch.linspace(-1, 1, t.numel(), device=dev).view(t.shape)).sum()
            enc = out["encoded_spconv_tensor"]
            loss = loss + (enc.features ** 2).mean()
            loss.backward()
        finally:
            if ctx:
                ctx.__exit__(None, None, None)
        res[dev] = (out, f.grad, {n: p.grad for n, p in m.named_parameters()}, {n: b for n, b in m.named_buffers() if b.is_floating_point()})
    for name in ("x_conv1", "x_conv2", "x_conv3", "x_conv4"):
        a, b = res["cuda"][0]["multi_scale_3d_features"][name], res["cpu"][0]["multi_scale_3d_features"][name]
        assert a.spatial_shape == b.spatial_shape and torch.equal(a.indices.cpu(), b.indices), name
        close(a.features, b.features, 2e-4, name)
        assert res["cuda"][0]["multi_scale_3d_strides"][name] == {"x_conv1": 1, "x_conv2": 2, "x_conv3": 4, "x_conv4": 8}[name]
    a, b = res["cuda"][0]["encoded_spconv_tensor"], res["cpu"][0]["encoded_spconv_tensor"]
    assert a.spatial_shape == b.spatial_shape == [2, 5, 6] and torch.equal(a.indices.cpu(), b.indices)
    close(a.features, b.features, 2e-4, "encoded")
    close(res["cuda"][1], res["cpu"][1], 2e-3, "d voxel_features")
    for n, gcpu in res["cpu"][2].items():
        close(res["cuda"][2][n], gcpu, 2e-3, "grad " + n)
    for n, bcpu in res["cpu"][3].items():
        close(res["cuda"][3][n], bcpu, 1e-4, "buffer " + n)


def test_voxel_query_through_hash_table_equals_dense_table(oracle):
    """mgar_voxel_query_hash_stack == mgar_voxel_query_stack == the oracle, on the same voxels (generate_voxel2pinds dense=False)."""
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import voxel_query_utils as vq
    from multimodal_gar_amd.pcdet.utils import common_utils
    from multimodal_gar_amd.pcdet.utils.spconv_utils import SparseConvTensor
    shape = (9, 40, 44)
    idx = sparse_sites(11, 2, shape, 0.15)
    vs, lo = np.array([0.25, 0.25, 0.5], np.float32), np.array([-5.0, -5.0, -2.0], np.float32)
    xyz = ((idx[:, [3, 2, 1]].numpy().astype(np.float32) + 0.5) * vs + lo).astype(np.float32)
    rng = np.random.default_rng(12)
    m = 600
    q = np.stack([rng.uniform(-5.5, 6.5, m), rng.uniform(-5.5, 5.5, m), rng.uniform(-2.5, 2.8, m)], 1).astype(np.float32)
    q[:25] += 30.0                                        # far outside the grid: empty neighbourhoods
    cells = np.floor((q - lo) / vs).astype(np.int32)
    bidx = rng.integers(0, 2, m).astype(np.int32)
    order = np.argsort(bidx, kind="stable")
    q, cells, bidx = q[order], cells[order], bidx[order]
    new_coords = np.concatenate([bidx[:, None], cells[:, [2, 1, 0]]], 1).astype(np.int32)      # [b, z, y, x]
    sp = SparseConvTensor(torch.zeros(idx.shape[0], 1).cuda(), idx.cuda(), list(shape), 2)
    dense = common_utils.generate_voxel2pinds(sp, dense=True)
    table = common_utils.generate_voxel2pinds(sp, dense=False)
    assert torch.is_tensor(dense) and dense.shape == (2,) + shape and not torch.is_tensor(table)
    args = (torch.from_numpy(xyz).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(new_coords).cuda())
    for rng_zyx, radius, ns in (((2, 3, 3), 0.9, 16), ((1, 9, 2), 1.3, 8), ((4, 4, 4), 0.6, 32)):
        a = vq.voxel_query_raw(rng_zyx, radius, ns, *args, dense)
        b = vq.voxel_query_raw(rng_zyx, radius, ns, *args, table)
        assert torch.equal(a, b)
        want = oracle.voxel_query(rng_zyx, radius, ns, xyz, q, new_coords, dense.cpu().numpy())    # raw kernel output of the C oracle
        # the oracle leaves the slots of an empty row (beyond its -1 marker) at the caller's zeros, as the kernels do
        assert np.array_equal(a.cpu().numpy(), want)
        assert (a[:, 0] >= 0).sum() > 50 and (a[:, 0] < 0).sum() > 5


@pytest.mark.parametrize("subm,kernel,stride,padding,cin,cout", [(True, 3, 1, 1, 16, 32), (False, 3, 2, 1, 32, 64), (False, (3, 1, 1), (2, 1, 1), 0, 64, 128),
                                                                  (True, 3, 1, 1, 4, 16)])
def test_pair_list_kernels_equal_the_table_driven_ones(subm, kernel, stride, padding, cin, cout, monkeypatch):
    """Forward, data gradient and weight gradient over the compacted pair lists (one launch per kernel offset, no atomics)
    against the table-driven gather-GEMM / chunked dW on a few thousand sites; twice the same result (fixed summation order)."""
    from multimodal_gar_amd import sparse_ops
    shape, batch = [12, 40, 44], 3
    idx = sparse_sites(11, batch, tuple(shape), 0.10).cuda()
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(idx.shape[0], cin, generator=g).cuda()
    kk = sparse_ops._triple(kernel)
    w = (torch.randn(cout, *kk, cin, generator=g) / (cin * kk[0] * kk[1] * kk[2]) ** 0.5).cuda()
    res = []
    for pairs in (True, False, True):
        monkeypatch.setattr(sparse_ops, "PAIRS_FORWARD", pairs)
        monkeypatch.setattr(sparse_ops, "PAIRS_DGRAD", pairs)
        monkeypatch.setattr(sparse_ops, "_pow2", (lambda v: v >= 1 and (v & (v - 1)) == 0) if pairs else (lambda v: False))
        f, ww = feats.clone().requires_grad_(True), w.clone().requires_grad_(True)
        out, oidx, _ = sparse_ops.sparse_conv3d(f, idx, shape, batch, ww, kernel, stride, padding, subm, {}, "k")
        cot = torch.linspace(-1, 1, out.numel(), device="cuda").view(out.shape)
        (out * cot).sum().backward()
        res.append((out.detach(), f.grad, ww.grad, oidx))
    assert idx.shape[0] > 3000 and torch.equal(res[0][3], res[1][3])
    for a, b, what in zip(res[0][:3], res[1][:3], ("out", "d features", "d weight")):
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-6), what
    for a, b in zip(res[0][:3], res[2][:3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("subm,kernel,stride,padding,n_scale", [(True, 3, 1, 1, 1), (False, 3, 2, 1, 1), (False, (3, 1, 1), (2, 1, 1), 0, 1),
                                                                 (True, 3, 1, 1, 6)])
def test_device_pair_list_builder_equals_the_torch_definition(subm, kernel, stride, padding, n_scale):
    """csrc/sparse_conv.hip sp_pairs_{count,scan,fill}: the pair lists of a rulebook (per offset: the (input row, output row) of every
    output site with a neighbour, ascending output row) against their definition in torch (mask -> nonzero -> gather); more
    than one 1 024-row block, ragged tail, empty offsets."""
    from multimodal_gar_amd import sparse_ops
    shape, batch = [12, 40 * n_scale, 44], 3
    idx = sparse_sites(5, batch, tuple(shape), 0.10).cuda()
    rb = sparse_ops.Rulebook(idx, shape, batch, kernel, stride, padding, subm)
    pair_i, pair_o, items_dw, start_dw, n_dw, items_fw, start_fw = rb.pairs()
    mask_t = (rb.nbr >= 0).t().contiguous()
    counts = mask_t.sum(1)
    ko = torch.nonzero(mask_t)
    want_o = ko[:, 1].int()
    want_i = rb.nbr.t()[mask_t]
    p = int(counts.sum())
    assert p == rb.pair_count() and p > 1000 and rb.nbr.shape[0] > (1024 if n_scale == 1 else 4096)
    assert torch.equal(pair_o[:p], want_o) and torch.equal(pair_i[:p], want_i)
    # the items cover every pair exactly once, offset by offset
    it = items_dw[:n_dw].cpu().numpy()
    covered = sum(int(e - b) for _, b, e, _ in it)
    assert covered == p
    offs = np.concatenate([[0], np.cumsum(counts.cpu().numpy())])
    for k, b, e, _ in it:
        assert offs[k] <= b < e <= offs[k + 1]


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 64), (24, 40)])
@pytest.mark.parametrize("flip", [0, 1])
def test_register_gather_kernel_equals_lds_kernel(cin, cout, flip):
    """spconv_os_kernel (gathered rows straight into the MFMA's A registers, double-buffered W in LDS; round 3) against the
    LDS-staged spconv_gather_gemm_kernel on the same table: same sums in another order.  Ragged last tile, rows without any
    neighbour, mirrored offsets (the submanifold data gradient); (24, 40) falls outside the register kernel's channel table and
    must take the LDS kernel either way."""
    from multimodal_gar_amd import _lib as L, sparse_ops
    shape, batch = [10, 36, 40], 2
    idx = sparse_sites(3, batch, tuple(shape), 0.12).cuda()
    rb = sparse_ops.Rulebook(idx, shape, batch, 3, 1, 1, True)
    n = idx.shape[0]
    assert n % 128 != 0 and n > 2000
    g = torch.Generator().manual_seed(cin * 7 + cout)
    feats = torch.randn(n, cin, generator=g).cuda()
    w = (torch.randn(27, cin, cout, generator=g) / (27 * cin) ** 0.5).cuda()
    outs = []
    for on in (1, 0):
        L.call("mgar_spconv_set_register_gather", on)
        try:
            outs.append(sparse_ops._gather_gemm(n, 27, cin, cout, feats, rb.nbr, w, flip))
        finally:
            L.call("mgar_spconv_set_register_gather", 1)
    a, b = outs
    assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-6)
    # and against the definition in float64
    nb = rb.nbr.long()
    want = torch.zeros(n, cout, dtype=torch.float64, device="cuda")
    for k in range(27):
        wk = w[26 - k if flip else k].double()
        has = nb[:, k] >= 0
        want[has] += feats[nb[has, k]].double() @ wk
    assert (a.double() - want).abs().max().item() <= 2e-5 * (want.abs().max().item() + 1e-6)


def test_rulebook_rejects_duplicate_coordinates_when_asked():
    """the sparse kernels assume unique voxel coordinates (as spconv does); Rulebook(check_unique=True) asserts it."""
    from multimodal_gar_amd.sparse_ops import Rulebook
    idx = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 4], [0, 1, 2, 3]], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        Rulebook(idx, (4, 8, 8), 1, 3, 1, 1, True, check_unique=True)
    Rulebook(idx[:2].contiguous(), (4, 8, 8), 1, 3, 1, 1, True, check_unique=True)
