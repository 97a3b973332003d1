"""Micro-benchmark of csrc/input_prep.hip at the shipped sizes (Multimodal_cfg/mil3.yaml): one clip of 15 stitched JRDB frames
480 x 3760 -> 720 x 1280 float32, and one key-frame cloud pair; prints one JSON line per kernel with the achieved HBM rate
(algorithmic bytes: source bytes once + output once) and the host (Pillow + float32 normalisation) time for the same clip.

    python tools/bench_input_prep.py [--host | --profile]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from multimodal_gar_amd import input_ops  # noqa: E402
from multimodal_gar_amd.data.utils import jrdb_transforms as jt  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    rng = np.random.default_rng(0)
    t, ih, iw, oh, ow = 15, 480, 3760, 720, 1280
    host = rng.integers(0, 256, (t, ih, iw, 3), dtype=np.uint8)
    frames = torch.from_numpy(host).cuda()
    if "--profile" in sys.argv:                                   # a handful of launches for rocprofv3 (--kernel-trace / --pmc)
        out = torch.empty((3, t, oh, ow), dtype=torch.float32, device="cuda")
        for _ in range(5):
            input_ops.resize_normalize(frames, (oh, ow), layout="cthw", out=out)
        torch.cuda.synchronize()
        return
    for dtype, name, ob in ((torch.float32, "f32", 4), (torch.bfloat16, "bf16", 2)):
        out = torch.empty((3, t, oh, ow), dtype=dtype, device="cuda")
        ms = timed(lambda: input_ops.resize_normalize(frames, (oh, ow), dtype=dtype, layout="cthw", out=out))
        by = t * (ih * iw * 3 + oh * ow * 3 * ob)
        print(json.dumps({"kernel": "image_resize_normalize_kernel", "out": name, "clip": [t, ih, iw, oh, ow], "ms": round(ms, 4),
                          "algorithmic_MB": round(by / 1e6, 1), "GB/s": round(by / ms / 1e6, 1), "frac_of_8TB/s": round(by / ms / 1e6 / 8000, 3)}))
    up = torch.from_numpy((rng.normal(size=(131072, 4)) * 30).astype(np.float32)).cuda()
    lo = torch.from_numpy((rng.normal(size=(131072, 4)) * 30).astype(np.float32)).cuda()
    tu, tl = jt.rigid_transform("upper"), jt.rigid_transform("lower")
    ms = timed(lambda: input_ops.velodyne_merge_crop(up, lo, tu, tl, [-100, -100, -25, 100, 100, 25]))
    print(json.dumps({"kernel": "velodyne_merge_crop (3 launches + count read-back)", "points": 262144, "ms": round(ms, 4)}))
    h2d = torch.from_numpy(host).pin_memory()
    ms = timed(lambda: h2d.cuda(non_blocking=True), reps=10)
    print(json.dumps({"step": "upload of the uint8 clip (pinned)", "MB": round(host.nbytes / 1e6, 1), "ms": round(ms, 3)}))
    if "--host" in sys.argv:
        from PIL import Image
        from multimodal_gar_amd.dataloader import resize_to_tensor_normalize
        t0 = time.time()
        torch.stack([resize_to_tensor_normalize(Image.fromarray(f), (oh, ow)) for f in host])
        print(json.dumps({"step": "host: Pillow resize + float32 normalise, same clip, 1 thread", "ms": round((time.time() - t0) * 1e3, 1)}))


if __name__ == "__main__":
    main()
