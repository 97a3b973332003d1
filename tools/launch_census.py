#!/usr/bin/env python3
"""Where do the launches of one step come from?  (one clip per rank = the 8-GPU point of config c4: ~1 900 launches, 40 ms.)

Runs ONE eager training step of the clip model and counts, per forward module bucket (first `--depth` components of the module
path) and per backward autograd node, (a) the aten ops that launch device work and (b) the launches of this library's kernels
(csrc, through multimodal_gar_amd._lib.call).  Output: a table on stdout, largest first.

    python tools/launch_census.py --clips 1 [--depth 4] [--route pointnet2]
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multimodal_gar_amd  # noqa: E402,F401
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from multimodal_gar_amd import _lib as L, workload as W  # noqa: E402
from multimodal_gar_amd.op_timer import _NO_KERNEL  # noqa: E402


class Census(TorchDispatchMode):
    def __init__(self, names, depth):
        super().__init__()
        self.names, self.depth = names, depth
        self.stack = []
        self.aten = collections.Counter()     # (bucket, op) -> launches
        self.lib = collections.Counter()      # (bucket, entry point) -> calls
        self.phase = "forward"

    def bucket(self):
        if self.phase != "forward":
            node = torch._C._current_autograd_node() if hasattr(torch._C, "_current_autograd_node") else None
            return "%s:%s" % (self.phase, node.name() if node is not None else "-")
        return ".".join(self.stack[-1].split(".")[:self.depth]) if self.stack else "(top)"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if not getattr(func, "is_view", False) and name not in _NO_KERNEL:
            self.aten[(self.bucket(), name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=1)
    ap.add_argument("--frames", type=int, default=15)
    ap.add_argument("--actors", type=int, default=32)
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--route", default="pointnet2")
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
    step = W.TrainStep(args.actors, args.points, dev, route=args.route)
    step.module.overlap_branches = False
    batch = W.make_batch(100, args.clips, args.frames, args.actors, args.points, args.height, args.width, dev)
    step.run_eager(batch)
    step.run_eager(batch)
    torch.cuda.synchronize()
    names = {id(m): n for n, m in step.module.named_modules()}
    census = Census(names, args.depth)
    def enter(m, a):
        census.stack.append(names.get(id(m), type(m).__name__))

    def leave(m, a, o):     # must return None: a returned value would replace the module's output
        if census.stack:
            census.stack.pop()
    pre = torch.nn.modules.module.register_module_forward_pre_hook(enter)
    post = torch.nn.modules.module.register_module_forward_hook(leave)
    real_call = L.call

    def counting_call(name, *a):
        census.lib[(census.bucket(), name)] += 1
        return real_call(name, *a)
    L.call = counting_call
    try:
        with census:
            step.opt.zero_grad(set_to_none=True)
            out = step.model(batch)
            census.phase = "loss"
            loss = step._loss_of(out, batch)
            census.phase = "backward"
            loss.backward()
            census.phase = "optimizer"
            step.opt.step()
        torch.cuda.synchronize()
    finally:
        L.call = real_call
        pre.remove(); post.remove()
    per_bucket = collections.Counter()
    for (b, _), n in list(census.aten.items()) + list(census.lib.items()):
        per_bucket[b] += n
    total = sum(per_bucket.values())
    print("one eager step, %d clip(s): %d launching calls (%d aten ops, %d library entry-point calls)"
          % (args.clips, total, sum(census.aten.values()), sum(census.lib.values())))
    for b, n in per_bucket.most_common(args.top):
        ops = collections.Counter()
        for (bb, op), k in census.aten.items():
            if bb == b:
                ops["aten." + op] += k
        for (bb, op), k in census.lib.items():
            if bb == b:
                ops[op.replace("mgar_", "hip:")] += k
        print("%6d  %-64s %s" % (n, b[:64], ", ".join("%s x%d" % kv for kv in ops.most_common(6))))


if __name__ == "__main__":
    main()
