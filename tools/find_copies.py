"""Diagnostic: where do the large aten.clone / aten.cat / aten.copy_ launches of one eager c3 train step come from?
Prints, per (op, shapes), the innermost frames inside this repository."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_gar_amd import workload as W  # noqa: E402


OPS = ("clone", "cat", "copy_", "_to_copy", "contiguous") + tuple(os.environ.get("MGAR_SPY_OPS", "").split(",")) if os.environ.get("MGAR_SPY_OPS") else ("clone", "cat", "copy_", "_to_copy", "contiguous")


class Spy(TorchDispatchMode):
    def __init__(self, min_numel):
        super().__init__()
        self.min_numel, self.seen = min_numel, collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in OPS and torch.is_tensor(out) and max(out.numel(), max([a.numel() for a in args if torch.is_tensor(a)] or [0])) >= self.min_numel:
            frames = [f for f in traceback.extract_stack() if ROOT in f.filename and "find_copies" not in f.filename][-3:]
            where = " <- ".join("%s:%d" % (os.path.relpath(f.filename, ROOT), f.lineno) for f in reversed(frames))
            self.seen[(name, tuple(out.shape), where)] += 1
        return out


def main():
    dev = torch.device("cuda")
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    step = W.TrainStep(32, 16384, dev, seed=1, manual_allreduce=True)
    batch = W.make_batch(3, clips, 15, 32, 16384, 720, 1280, dev)
    step.run_eager(batch) if hasattr(step, "run_eager") else step.step(batch)
    torch.cuda.synchronize()
    with Spy(8_000_000) as spy:
        step.run_eager(batch) if hasattr(step, "run_eager") else step.step(batch)
    torch.cuda.synchronize()
    for (name, shape, where), n in sorted(spy.seen.items(), key=lambda kv: -kv[0][1].__len__()):
        numel = 1
        for s in shape:
            numel *= s
        print("%-8s %-26s x%d  %7.1f MB  %s" % (name, "x".join(map(str, shape)), n, numel * 4 / 1e6, where))


if __name__ == "__main__":
    main()
