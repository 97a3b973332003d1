"""Seeded inputs of the train_utils fixtures (shared by the generator, which runs the REFERENCE's functions, and the tests):
ids with -1 padding as dataloader.py:245-253 pads them, one-hot-ish action labels (B, MAX, 27), sigmoid-range A_theta."""
import numpy as np
import torch

MAX = 12


def make_case(seed, batch=3):
    rng = np.random.default_rng(seed)
    person_id = -np.ones((batch, MAX), np.int64)
    group_id = -np.ones((batch, MAX), np.int64)
    for b in range(batch):
        n = int(rng.integers(2, MAX - 1))
        person_id[b, :n] = rng.permutation(50)[:n]
        group_id[b, :n] = rng.integers(0, max(2, n // 2), n)
    action = (rng.random((batch, MAX, 27)) < 0.25).astype(np.float32)
    sg_activity = (rng.random((batch, MAX, 27)) < 0.25).astype(np.float32)
    g = torch.Generator().manual_seed(seed)
    return {"person_id": torch.from_numpy(person_id), "social_group_id": torch.from_numpy(group_id),
            "action": torch.from_numpy(action), "social_group_activity": torch.from_numpy(sg_activity),
            "A_theta": torch.rand((batch, MAX, MAX), generator=g) * 0.98 + 0.01}
