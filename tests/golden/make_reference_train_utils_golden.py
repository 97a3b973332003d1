"""Generates tests/golden/reference_train_utils.npz by RUNNING THE REFERENCE'S OWN train_utils.py functions (imported from
/root/reference in this container only; the file imports nothing but torch and time):
    get_num_person, get_num_social_group, get_adjacency, get_laplacian, get_eig_loss2, get_label_from_action,
    sid2AdjMat, Adj2Deg, Adj2Lap
get_adjacency ends with ``.cuda()`` on each matrix (train_utils.py:109); there is no GPU here, so Tensor.cuda is made the
identity for the duration of this script.  train_func.py (the script that holds the loss composition) is never imported:
it runs at import and opens a network session.
Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_train_utils_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
from train_cases import make_case  # noqa: E402


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, "/root/reference")
    import train_utils as R
    out = {}
    for seed in (1, 2, 3):
        c = make_case(seed)
        tag = "case%d/" % seed
        pn = R.get_num_person(c["person_id"])
        out[tag + "person_num"] = np.array(pn)
        out[tag + "social_group_num"] = np.array(R.get_num_social_group(c["social_group_id"]))
        A_hat = R.get_adjacency(c["social_group_id"], pn)
        labels = R.get_label_from_action(c["action"], pn)
        for b in range(len(pn)):
            out[tag + "A_hat%d" % b] = A_hat[b].numpy()
            out[tag + "lap%d" % b] = R.get_laplacian(A_hat[b]).numpy()
            out[tag + "sid2adj%d" % b] = R.sid2AdjMat(c["social_group_id"][b]).numpy()
            for k in range(7):
                out[tag + "label%d_%d" % (k, b)] = labels[k][b].numpy()
        A_theta = [c["A_theta"][b, :pn[b], :pn[b]] for b in range(len(pn))]
        # on a GPU ``zeros(requires_grad=True).to(device)`` is a non-leaf copy; on CPU it stays the leaf itself and the
        # reference's in-place ``eig_loss += ...`` (:141) is refused by autograd -- evaluated without autograd here (same value)
        with torch.no_grad():
            out[tag + "eig_loss2"] = R.get_eig_loss2(A_theta, A_hat).detach().numpy()
        A = torch.stack([torch.nn.functional.pad(a, (0, 12 - a.shape[0], 0, 12 - a.shape[0])) for a in A_hat])
        out[tag + "adj2deg"] = R.Adj2Deg(A).numpy()
        out[tag + "adj2lap"] = R.Adj2Lap(A).numpy()
    path = os.path.join(HERE, "reference_train_utils.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
