"""Deterministic, construction-order independent parameter fill used by the reference-golden
fixtures: every state-dict entry is drawn from a generator seeded by its NAME, so the reference
module and this repo's module receive identical weights without storing them (the real models
are 7.5-11 M parameters; the fixtures hold inputs and outputs only)."""
import zlib

import torch


def fill_deterministic(module, seed=0):
    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd):
            t = sd[name]
            if not t.is_floating_point():
                continue  # num_batches_tracked
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) % (2 ** 31))
            if name.endswith("running_var"):
                v = torch.rand(t.shape, generator=g) + 0.5
            elif name.endswith("running_mean"):
                v = torch.randn(t.shape, generator=g) * 0.1
            elif t.dim() <= 1:
                # biases / norm scales: keep normalisation layers near identity but not trivial
                v = torch.randn(t.shape, generator=g) * 0.1 + (1.0 if name.endswith("weight") else 0.0)
            else:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) / max(fan_in, 1) ** 0.5
            t.copy_(v.to(t.dtype))
    return module
