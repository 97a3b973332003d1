"""Generates tests/golden/reference_pcdet_modules.npz by RUNNING THE REFERENCE'S OWN Python op modules
(imported from /root/reference in this container only) with the C oracle standing in for the CUDA
extension they bind:

    pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py   PointnetSAModuleMSG, PointnetFPModule
    pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py   StackSAModuleMSG, StackPointnetFPModule
    pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py  NeighborVoxelSAModuleMSG
    (and, through them, pointnet2_utils.py / voxel_query_utils.py: the autograd Functions, QueryAndGroup,
     VoxelQueryAndGrouping -- SURVEY.md section 8 rows a8, a9, a12, a14, a15)

This pins the COMPOSITION (which op feeds which, the -1 / empty-ball handling, the channel order of the
concat, the index re-basing of voxel_query_utils.py:83-91, BN placement, pooling) to the reference's own
code; the kernels underneath are oracle/mgar_oracle.c (their parity stays "unpinned": the reference ships
no vectors and its CUDA cannot be built here).

How the import works: the two op directories are mounted as packages `refops_batch` / `refops_stack`
(module objects whose __path__ is the reference directory), with `pointnet2_batch_cuda` /
`pointnet2_stack_cuda` pre-registered as stub modules whose functions are oracle/cpu_backend.py's.  The
reference's wrappers allocate with torch.cuda.IntTensor / FloatTensor; those two names are pointed at the
CPU tensor types for the duration of this script.  Nothing else under /root/reference is executed.

Weights are NOT stored: both sides fill them with tests/golden/param_fill.py.
Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_module_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from oracle import cpu_backend as CB  # noqa: E402
from oracle import oracle as O  # noqa: E402
from param_fill import fill_deterministic  # noqa: E402
from module_cases import CASES, make_inputs  # noqa: E402

REF_OPS = "/root/reference/pcdet/ops/pointnet2"


def mount(pkg, path, ext_name, impl):
    m = types.ModuleType(pkg)
    m.__path__ = [path]
    sys.modules[pkg] = m
    ext = types.ModuleType(pkg + "." + ext_name)
    for name in dir(impl):
        if name.endswith("_wrapper"):
            setattr(ext, name, getattr(impl, name))
    sys.modules[pkg + "." + ext_name] = ext
    setattr(m, ext_name, ext)
    return m


def main():
    O.build()
    torch.cuda.IntTensor = torch.IntTensor        # the reference's wrappers allocate with these names
    torch.cuda.FloatTensor = torch.FloatTensor
    mount("refops_batch", os.path.join(REF_OPS, "pointnet2_batch"), "pointnet2_batch_cuda", CB._Batch)
    mount("refops_stack", os.path.join(REF_OPS, "pointnet2_stack"), "pointnet2_stack_cuda", CB._Stack)
    mods = {
        "batch": importlib.import_module("refops_batch.pointnet2_modules"),
        "stack": importlib.import_module("refops_stack.pointnet2_modules"),
        "voxel": importlib.import_module("refops_stack.voxel_pool_modules"),
    }
    out = {}
    for case in CASES:
        cls = getattr(mods[case["where"]], case["cls"])
        m = fill_deterministic(cls(**case["kwargs"]()), seed=case["seed"]).train()
        ins = make_inputs(case)
        args = [t.clone().requires_grad_(True) if rg else t for t, rg in ins]
        res = m(*args)
        outs = [o for o in (res if isinstance(res, (tuple, list)) else (res,)) if torch.is_tensor(o) and o.is_floating_point()]
        y = outs[-1]
        # a fixed, non-uniform cotangent so that the backward exercises every output element differently
        cot = torch.linspace(-1.0, 1.0, y.numel()).view(y.shape)
        (y * cot).sum().backward()
        tag = case["name"]
        out[tag + "/y"] = y.detach().numpy()
        for i, (a, (_, rg)) in enumerate(zip(args, ins)):
            if rg:
                out["%s/grad_in%d" % (tag, i)] = a.grad.numpy()
        for n, p in m.named_parameters():
            out["%s/grad_param/%s" % (tag, n)] = p.grad.numpy()
        for n, b in m.named_buffers():
            if b.is_floating_point():
                out["%s/buffer/%s" % (tag, n)] = b.detach().numpy()
        print("%-28s y %s  |y| %.4f  params %d" % (tag, tuple(y.shape), float(y.abs().mean()), sum(p.numel() for p in m.parameters())))
    path = os.path.join(HERE, "reference_pcdet_modules.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
