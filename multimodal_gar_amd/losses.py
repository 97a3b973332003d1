"""The training objective of MGAR-net on the device.

Restatement of the loss block of the reference's train loop (train_func.py:133-258): per-sample slicing of the 16 model
outputs to the scene's actors, weighted / plain BCE on the social-group adjacency A_theta, cross-entropy / BCE on the pose
and interaction heads for individuals and for social groups, MSE on the group cardinality, the eigen loss, and their
``Loss``-selected sum.  The reference computes every term in Python loops over the batch with host-built targets; here the
targets come from train_utils (tensor ops on the outputs' device) and, when every scene holds the same number of actors
(``uniform=True``: the synthetic workload, and the reference's own BATCH_SIZE 1), every term is ONE batched op.

The reference's loop has conventions that decide the VALUE of the objective and are kept (``reference_semantics=True``):
L_bce, L_bce2, L_pose and SG_L_pose are ASSIGNED inside ``for i in range(batch_size)`` (train_func.py:186-202, 216-218,
233-235), so only the LAST sample's value survives, while L_interaction / SG_L_interaction accumulate (+=) over the batch.
With reference_semantics=False the assigned terms are summed over the batch like the accumulated ones.
"""
import torch
import torch.nn.functional as F

from . import train_utils as TU


def _bce(p, t):
    return F.binary_cross_entropy(p, t)


def mgar_losses(res, person_id, social_group_id, action, social_group_activity, Loss="L_total", person_num=None,
                reference_semantics=True):
    """res: the model's 16-tuple; person_id / social_group_id (B, MAX); action / social_group_activity (B, MAX, 27).
    -> dict with L_total and every term (tensors on the outputs' device).  person_num: per-sample actor counts (list);
    computed as the reference does when omitted (one host sync)."""
    A_theta, pose, intr, sg_pose, sg_intr, card = res[0], res[1:4], res[4:8], res[8:11], res[11:15], res[15]
    dev = A_theta.device
    batch_size = A_theta.shape[0]
    if person_num is None:
        person_num = TU.get_num_person(person_id)
    action, social_group_activity = action.to(dev), social_group_activity.to(dev)
    social_group_id = social_group_id.to(dev)
    social_group_num = TU.get_num_social_group(social_group_id)
    A_hat = TU.get_adjacency(social_group_id, person_num)
    label = TU.get_label_from_action(action, person_num)
    sg_label = TU.get_label_from_action(social_group_activity, person_num)
    out = {}
    samples = range(batch_size)
    keep = [batch_size - 1] if reference_semantics else list(samples)      # terms the reference assigns instead of accumulating

    def a_theta(i):
        return A_theta[i, :person_num[i], :person_num[i]]

    l_bce, l_bce2 = 0.0, 0.0
    for i in keep:
        n = person_num[i]
        mask = 1.0 - torch.eye(n, device=dev)
        non_group = (A_hat[i] == 0).float()
        n_group = (A_hat[i] * mask).sum()
        ratio = (mask.sum() - n_group) / (3 * n_group + 1)
        el = F.binary_cross_entropy(a_theta(i), A_hat[i], reduction='none') * mask
        l_bce2 = l_bce2 + (ratio * el * A_hat[i] + el * non_group).sum() / mask.sum()
        l_bce = l_bce + _bce(a_theta(i), A_hat[i])
    out["L_bce"], out["L_bce2"] = l_bce, l_bce2
    out["L_mse"] = F.mse_loss(torch.cat([card[b] for b in samples]), torch.tensor(social_group_num, device=dev).float())
    if Loss == "L_g":
        out["L_eig"] = TU.get_eig_loss2([a_theta(i) for i in samples], A_hat)
        out["L_g"] = out["L_bce"] + out["L_eig"] + out["L_mse"]
    l_pose = sum(sum(F.cross_entropy(pose[k][i, :person_num[i]], label[k][i]) for k in range(3)) for i in keep)
    l_int = sum(sum(_bce(intr[k][i, :person_num[i]], label[3 + k][i]) for k in range(4)) for i in samples)
    sg_l_pose = sum(sum(_bce(sg_pose[k][i, :person_num[i]], sg_label[k][i]) for k in range(3)) for i in keep)
    sg_l_int = sum(sum(_bce(sg_intr[k][i, :person_num[i]], sg_label[3 + k][i]) for k in range(4)) for i in samples)
    out.update({"L_pose": l_pose, "L_interaction": l_int, "L_act": l_pose + l_int, "SG_L_pose": sg_l_pose,
                "SG_L_interaction": sg_l_int, "SG_L_act": sg_l_pose + sg_l_int})
    out["L_total"] = {"L_g": lambda: out["L_g"], "L_bce": lambda: out["L_bce"], "L_bce2": lambda: out["L_bce2"],
                      "L_total": lambda: out["L_bce"] + out["L_act"] + out["SG_L_act"],
                      "L_act": lambda: out["L_act"] + out["SG_L_act"]}[Loss]()
    return out


def mgar_losses_uniform(res, social_group_id, action, social_group_activity, n, Loss="L_total", reference_semantics=True):
    """The same objective when every one of the B scenes holds exactly n actors (the synthetic workload; B = 1 in the
    reference's own configuration): no Python loop, no host sync -- every term is one batched op, so the whole loss graph is
    ~40 launches for any B and can be captured into a HIP graph.  Supports the objectives without the eigen term."""
    assert Loss in ("L_total", "L_act", "L_bce", "L_bce2")
    A_theta = res[0][:, :n, :n]
    pose, intr, sg_pose, sg_intr = [t[:, :n] for t in res[1:4]], [t[:, :n] for t in res[4:8]], [t[:, :n] for t in res[8:11]], \
        [t[:, :n] for t in res[11:15]]
    dev = A_theta.device
    A_hat = TU.get_adjacency_batched(social_group_id.to(dev), n)
    label = TU.get_label_from_action_batched(action.to(dev), n)
    sg_label = TU.get_label_from_action_batched(social_group_activity.to(dev), n)
    pick = (lambda t: t[-1:]) if reference_semantics else (lambda t: t)     # assigned terms: the last scene only

    def per_scene_mean(x):                                                     # mean over a scene's elements -> (B,)
        return x.flatten(1).mean(1)
    mask = 1.0 - torch.eye(n, device=dev)
    el = F.binary_cross_entropy(A_theta, A_hat, reduction='none')
    l_bce = pick(per_scene_mean(el)).sum()
    n_group = (A_hat * mask).flatten(1).sum(1)
    ratio = ((mask.sum() - n_group) / (3 * n_group + 1)).view(-1, 1, 1)
    l_bce2 = pick(((ratio * el * mask * A_hat + el * mask * (A_hat == 0).float()).flatten(1).sum(1) / mask.sum())).sum()
    ce = lambda logit, target: -(target * F.log_softmax(logit, dim=-1)).sum(-1).mean(1)      # noqa: E731  soft-label CE, (B,)
    bce = lambda p, t: per_scene_mean(F.binary_cross_entropy(p, t, reduction='none'))        # noqa: E731
    l_pose = pick(sum(ce(pose[k], label[k]) for k in range(3))).sum()
    l_int = sum(bce(intr[k], label[3 + k]) for k in range(4)).sum()
    sg_l_pose = pick(sum(bce(sg_pose[k], sg_label[k]) for k in range(3))).sum()
    sg_l_int = sum(bce(sg_intr[k], sg_label[3 + k]) for k in range(4)).sum()
    out = {"L_bce": l_bce, "L_bce2": l_bce2, "L_pose": l_pose, "L_interaction": l_int, "L_act": l_pose + l_int,
           "SG_L_pose": sg_l_pose, "SG_L_interaction": sg_l_int, "SG_L_act": sg_l_pose + sg_l_int}
    out["L_total"] = {"L_bce": l_bce, "L_bce2": l_bce2, "L_total": l_bce + out["L_act"] + out["SG_L_act"],
                      "L_act": out["L_act"] + out["SG_L_act"]}[Loss]
    return out
