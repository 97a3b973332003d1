"""Shared cases for the float64 ground-truth parity tests (tests/test_fp64_truth_cpu.py, tests/test_fp64_truth_gpu.py):
one op module, seeded inputs, and the three evaluations of it --

  hip    : the module on the device (HIP kernels through the C ABI), fp32
  oracle : the same module code on the CPU with the C oracle bound in (oracle/cpu_backend.py), fp32
  truth  : the composition restated in float64 (oracle/fp64_truth.py; integer decisions from the C oracle)

each returning (outputs, gradients wrt the float inputs that require grad, gradients wrt the parameters in
named_parameters() order).  Loss = sum(out * cot) with a fixed cotangent, so gradients are linear in the outputs."""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from param_fill import fill_deterministic  # noqa: E402


def _scene(seed, b, n):
    from multimodal_gar_amd import synthetic as S
    sc = S.scene_batch(seed, b, 4, n)
    return torch.from_numpy(np.ascontiguousarray(sc["points"][:, :, :3]))


def _cot(shape, seed=0):
    g = torch.Generator().manual_seed(1000 + seed)
    return torch.randn(shape, generator=g, dtype=torch.float64)


class Case:
    def __init__(self, name, make, inputs, call, truth):
        """inputs: list of (tensor, requires_grad); call(mod, *inputs) -> the float output tensor of the module;
        truth(mod64, *inputs64) -> the same in float64."""
        self.name, self.make, self.inputs, self.call, self.truth_fn = name, make, inputs, call, truth
        self.module = fill_deterministic(make(), seed=3).train()

    def _run(self, mod, ins, fn, dtype):
        out = fn(mod, *ins)
        (out.double() * _cot(out.shape).to(out.device)).sum().backward()
        gi = [t.grad.detach().double().cpu() for t in ins if torch.is_tensor(t) and t.requires_grad]
        gp = [p.grad.detach().double().cpu() for _, p in mod.named_parameters() if p.grad is not None]
        return out.detach().double().cpu(), gi, gp

    def _inputs(self, dev, dtype):
        out = []
        for t, rg in self.inputs:
            x = t.detach().clone()
            if x.is_floating_point():
                x = x.to(dtype) if rg else x           # coordinates stay fp32-valued (fp32 tensors; the truth widens them)
            out.append(x.to(dev).requires_grad_(bool(rg)))
        return out

    def hip(self):
        mod = copy.deepcopy(self.module).cuda()
        return self._run(mod, self._inputs("cuda", torch.float32), self.call, torch.float32)

    def oracle(self):
        from oracle.cpu_backend import use_cpu_oracle
        mod = copy.deepcopy(self.module)
        with use_cpu_oracle():
            return self._run(mod, self._inputs("cpu", torch.float32), self.call, torch.float32)

    def truth(self):
        from oracle import fp64_truth as T
        mod = T.double_copy(self.module)
        return self._run(mod, self._inputs("cpu", torch.float64), self.truth_fn, torch.float64)


def cases():
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_modules as MB
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as MS
    from oracle import fp64_truth as T
    out = []
    g = torch.Generator().manual_seed(7)
    xyz = _scene(1, 2, 1024)
    feats = torch.randn(2, 5, 1024, generator=g)
    out.append(Case("sa_msg_batch", lambda: MB.PointnetSAModuleMSG(npoint=128, radii=[0.8, 2.0], nsamples=[8, 16],
                                                                   mlps=[[5, 16, 32], [5, 16, 32]]),
                    [(xyz, False), (feats, True)], lambda m, x, f: m(x, f)[1], lambda m, x, f: T.sa_msg_batch(m, x, f)[1]))
    # wide input: the device takes the "project, then group" route
    feats40 = torch.randn(2, 40, 1024, generator=g)
    out.append(Case("sa_msg_batch_projected", lambda: MB.PointnetSAModuleMSG(npoint=100, radii=[0.9, 2.0], nsamples=[8, 16],
                                                                             mlps=[[40, 16, 32], [40, 24, 24]]),
                    [(xyz, False), (feats40, True)], lambda m, x, f: m(x, f)[1], lambda m, x, f: T.sa_msg_batch(m, x, f)[1]))
    known = xyz[:, :100].contiguous()
    kf, uf = torch.randn(2, 12, 100, generator=g), torch.randn(2, 7, 1024, generator=g)
    out.append(Case("fp_batch", lambda: MB.PointnetFPModule(mlp=[19, 32, 16]),
                    [(xyz, False), (known, False), (uf, True), (kf, True)], lambda m, *a: m(*a), T.fp_batch))
    sx = _scene(2, 2, 700).reshape(-1, 3)
    cnt = torch.tensor([700, 700], dtype=torch.int32)
    new_xyz = torch.cat([sx[:50], sx[700:760] + 0.05, torch.tensor([[900., 900., 900.]])])   # last query: empty ball
    ncnt = torch.tensor([50, 61], dtype=torch.int32)
    sf = torch.randn(1400, 9, generator=g)
    out.append(Case("sa_msg_stack", lambda: MS.StackSAModuleMSG(radii=[0.9, 2.5], nsamples=[8, 16], mlps=[[9, 16], [9, 24, 32]]),
                    [(sx, False), (cnt, False), (new_xyz, False), (ncnt, False), (sf, True)],
                    lambda m, *a: m(*a)[1], T.stack_sa_msg))
    sf40 = torch.randn(1400, 40, generator=g)
    out.append(Case("sa_msg_stack_projected", lambda: MS.StackSAModuleMSG(radii=[0.9, 2.5], nsamples=[8, 16], mlps=[[40, 16], [40, 24, 32]]),
                    [(sx, False), (cnt, False), (new_xyz, False), (ncnt, False), (sf40, True)],
                    lambda m, *a: m(*a)[1], T.stack_sa_msg))
    q = new_xyz[:110].contiguous()
    qcnt = torch.tensor([50, 60], dtype=torch.int32)
    kf2 = torch.randn(110, 6, generator=g)
    out.append(Case("fp_stack", lambda: MS.StackPointnetFPModule(mlp=[15, 20]),
                    [(sx, False), (cnt, False), (q, False), (qcnt, False), (sf, True), (kf2, True)],
                    lambda m, *a: m(*a), T.stack_fp))
    return out


def rel_err(a, b):
    """max |a - b| relative to max |b| (the tests' norm) and the per-element relative error at |b| >= 1e-2 max |b|."""
    scale = b.abs().max().item() + 1e-300
    d = (a - b).abs()
    big = b.abs() >= 1e-2 * scale
    per = (d[big] / b.abs()[big]).max().item() if big.any() else 0.0
    return d.max().item() / scale, per
