"""Grouping targets and the training objective.

* multimodal_gar_amd/train_utils.py vs the outputs of the REFERENCE's own train_utils.py functions
  (tests/golden/reference_train_utils.npz, made by tests/golden/make_reference_train_utils_golden.py);
* multimodal_gar_amd/losses.py vs the oracle's restatement of the loss section of train_func.py:133-258
  (oracle/train_objective.py: that file is a script that cannot be imported -- it runs at import and opens a network
  session), including its assign-instead-of-accumulate terms; and the batched no-loop version vs the looped one.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from train_cases import MAX, make_case  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "reference_train_utils.npz"))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_train_utils_match_reference_functions(seed):
    from multimodal_gar_amd import train_utils as TU
    c = make_case(seed)
    tag = "case%d/" % seed
    pn = TU.get_num_person(c["person_id"])
    assert pn == GOLD[tag + "person_num"].tolist()
    assert TU.get_num_social_group(c["social_group_id"]) == GOLD[tag + "social_group_num"].tolist()
    A_hat = TU.get_adjacency(c["social_group_id"], pn)
    labels = TU.get_label_from_action(c["action"], pn)
    for b in range(len(pn)):
        assert np.array_equal(A_hat[b].numpy(), GOLD[tag + "A_hat%d" % b])
        assert np.array_equal(TU.get_laplacian(A_hat[b]).numpy(), GOLD[tag + "lap%d" % b])
        assert np.array_equal(TU.sid2AdjMat(c["social_group_id"][b]).numpy(), GOLD[tag + "sid2adj%d" % b])
        for k in range(7):
            assert np.array_equal(labels[k][b].numpy(), GOLD[tag + "label%d_%d" % (k, b)]), (k, b)
    A_theta = [c["A_theta"][b, :pn[b], :pn[b]] for b in range(len(pn))]
    got = TU.get_eig_loss2(A_theta, A_hat).detach().numpy()
    assert np.allclose(got, GOLD[tag + "eig_loss2"], rtol=1e-9, atol=1e-12), (got, GOLD[tag + "eig_loss2"])
    A = torch.stack([torch.nn.functional.pad(a, (0, MAX - a.shape[0], 0, MAX - a.shape[0])) for a in A_hat])
    assert np.array_equal(TU.Adj2Deg(A).numpy(), GOLD[tag + "adj2deg"]) and np.array_equal(TU.Adj2Lap(A).numpy(), GOLD[tag + "adj2lap"])


def _fake_outputs(seed, batch, grad=False):
    g = torch.Generator().manual_seed(seed)
    sig = lambda *s: torch.rand(*s, generator=g) * 0.98 + 0.01          # noqa: E731  sigmoid-range head outputs
    res = [sig(batch, MAX, MAX)] + [torch.randn(batch, MAX, 4, generator=g) for _ in range(3)] \
        + [sig(batch, MAX, k) for k in (2, 4, 7, 5)] + [sig(batch, MAX, 4) for _ in range(3)] + [sig(batch, MAX, k) for k in (2, 4, 7, 5)] \
        + [torch.rand(batch, 1, generator=g) * 4]
    return [t.requires_grad_(grad) for t in res]


@pytest.mark.parametrize("seed", [1, 2])
def test_losses_match_the_reference_loop(seed):
    from multimodal_gar_amd import losses, train_utils as TU
    c = make_case(seed)
    res = _fake_outputs(seed + 10, 3, grad=True)
    from oracle.train_objective import reference_losses
    want = reference_losses(res, c["person_id"], c["social_group_id"], c["action"], c["social_group_activity"], TU)
    got = losses.mgar_losses(res, c["person_id"], c["social_group_id"], c["action"], c["social_group_activity"], Loss="L_total")
    for k, v in want.items():
        assert torch.allclose(torch.as_tensor(got[k]).float(), torch.as_tensor(v).float(), rtol=1e-6, atol=1e-7), k
    gw = torch.autograd.grad(want["L_total"], res[:15], allow_unused=True)
    gg = torch.autograd.grad(got["L_total"], res[:15], allow_unused=True)
    for a, b in zip(gg, gw):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-8)
    g2 = losses.mgar_losses(res, c["person_id"], c["social_group_id"], c["action"], c["social_group_activity"], Loss="L_g")
    assert torch.isfinite(g2["L_total"]).all() and "L_eig" in g2


@pytest.mark.parametrize("ref_sem", [True, False])
def test_batched_losses_equal_looped_losses_for_uniform_actor_counts(ref_sem):
    from multimodal_gar_amd import losses
    batch, n = 5, 7
    rng = np.random.default_rng(4)
    pid = -np.ones((batch, MAX), np.int64); gid = -np.ones((batch, MAX), np.int64)
    for b in range(batch):
        pid[b, :n] = rng.permutation(40)[:n]
        gid[b, :n] = rng.integers(0, 3, n)
        gid[b, 0], gid[b, 1], gid[b, 2] = 0, 1, 2                            # 3 groups in every scene
    action = torch.from_numpy((rng.random((batch, MAX, 27)) < 0.3).astype(np.float32))
    sga = torch.from_numpy((rng.random((batch, MAX, 27)) < 0.3).astype(np.float32))
    res = _fake_outputs(21, batch, grad=True)
    looped = losses.mgar_losses(res, torch.from_numpy(pid), torch.from_numpy(gid), action, sga, Loss="L_total", reference_semantics=ref_sem)
    batched = losses.mgar_losses_uniform(res, torch.from_numpy(gid), action, sga, n, Loss="L_total", reference_semantics=ref_sem)
    for k in ("L_bce", "L_bce2", "L_pose", "L_interaction", "SG_L_pose", "SG_L_interaction", "L_total"):
        assert torch.allclose(torch.as_tensor(batched[k]), torch.as_tensor(looped[k]), rtol=1e-5, atol=1e-7), k
    ga = torch.autograd.grad(looped["L_total"], res[:15], allow_unused=True)
    gb = torch.autograd.grad(batched["L_total"], res[:15], allow_unused=True)
    for a, b in zip(gb, ga):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-8)
