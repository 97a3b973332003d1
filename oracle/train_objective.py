"""TEST INFRASTRUCTURE -- CPU restatement of the reference's training objective, never imported by the product.

Follows /root/reference/train_func.py:133-258 (the per-batch loss section of ``train_one_epoch``).  That file is a script:
it executes at import and opens a network session (SURVEY.md section 8c "must never be run"), so its loss section cannot be
called; this module restates it.  What is kept, because the product's ``losses.mgar_losses`` has to reproduce it:

* scenes are cut to their real actor count (person_num) before any loss (train_func.py:137-160);
* the adjacency terms (``L_bce``, ``L_bce2``, train_func.py:176-193) and the two pose terms (``L_pose`` :203-207,
  ``SG_L_pose`` :221-225) are ASSIGNED inside the loop over scenes -- only the LAST scene of the batch survives --
  while the interaction terms (:209-216, :227-235) are accumulated over scenes;
* the weighted adjacency BCE: off-diagonal mask, ratio = (#off-diagonal - #in-group) / (3 #in-group + 1) on in-group
  entries, weight 1 on the others, normalised by the number of off-diagonal entries (:178-190);
* ``L_mse`` on the cardinality head against the number of social groups (:195-197), not part of ``L_total``;
* ``L_total = L_bce + (L_pose + L_interaction) + (SG_L_pose + SG_L_interaction)`` (:246-247, Loss == "L_total").

Targets come from ``train_utils`` (get_num_person / get_num_social_group / get_adjacency / get_label_from_action), which
tests/golden/reference_train_utils.npz pins to the reference's own train_utils.py.
"""
import torch
import torch.nn.functional as F


def _per_scene(t, counts):
    return [t[b, :n] for b, n in enumerate(counts)]


def _weighted_adjacency_bce(a_theta, a_hat):
    """train_func.py:178-190 for one scene."""
    n = a_theta.shape[0]
    off_diag = 1.0 - torch.eye(n, dtype=a_theta.dtype)
    elementwise = F.binary_cross_entropy(a_theta, a_hat, reduction="none") * off_diag
    in_group = (a_hat * off_diag).sum()
    ratio = (off_diag.sum() - in_group) / (3 * in_group + 1)
    weighted = ratio * elementwise * a_hat + elementwise * (a_hat == 0)
    return weighted.sum() / off_diag.sum()


def reference_losses(outputs, person_id, social_group_id, action, social_group_activity, TU):
    """-> dict of the scalars train_func.py:176-247 forms for one batch (Loss == "L_total").  ``outputs`` is the model's
    16-tuple (gat_model.py:1696 order); ``TU`` the train_utils module that supplies the targets."""
    a_theta, poses, inter = outputs[0], outputs[1:4], outputs[4:8]
    sg_poses, sg_inter, card = outputs[8:11], outputs[11:15], outputs[15]
    counts = TU.get_num_person(person_id)
    scenes = range(a_theta.shape[0])
    a_theta_s = [a_theta[b, :n, :n] for b, n in enumerate(counts)]
    a_hat_s = TU.get_adjacency(social_group_id, counts)
    lab = TU.get_label_from_action(action, counts)                       # 3 pose targets (class ids) + 4 interaction targets
    sg_lab = TU.get_label_from_action(social_group_activity, counts)     # 7 multi-hot targets
    poses, inter = [_per_scene(t, counts) for t in poses], [_per_scene(t, counts) for t in inter]
    sg_poses, sg_inter = [_per_scene(t, counts) for t in sg_poses], [_per_scene(t, counts) for t in sg_inter]

    out = {}
    for b in scenes:                                                     # assigned, not accumulated (:176-193)
        out["L_bce2"] = _weighted_adjacency_bce(a_theta_s[b], a_hat_s[b])
        out["L_bce"] = F.binary_cross_entropy(a_theta_s[b], a_hat_s[b])
    groups = torch.tensor(TU.get_num_social_group(social_group_id)).float()
    out["L_mse"] = F.mse_loss(torch.cat([card[b] for b in scenes]), groups)
    for b in scenes:                                                     # assigned (:203-207)
        out["L_pose"] = sum(F.cross_entropy(poses[k][b], lab[k][b]) for k in range(3))
    out["L_interaction"] = sum(F.binary_cross_entropy(inter[k][b], lab[3 + k][b]) for b in scenes for k in range(4))
    for b in scenes:                                                     # assigned (:221-225)
        out["SG_L_pose"] = sum(F.binary_cross_entropy(sg_poses[k][b], sg_lab[k][b]) for k in range(3))
    out["SG_L_interaction"] = sum(F.binary_cross_entropy(sg_inter[k][b], sg_lab[3 + k][b]) for b in scenes for k in range(4))
    out["L_total"] = out["L_bce"] + (out["L_pose"] + out["L_interaction"]) + (out["SG_L_pose"] + out["SG_L_interaction"])
    return out
