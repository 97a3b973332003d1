"""Writes tests/golden/pil_resize.npz: seeded uint8 images and what Pillow's ``Image.resize(size, BILINEAR)`` -- the call
behind the reference's ``transforms.Resize`` (dataloader.py:47) -- returns for them, so that the oracle restatement and the
HIP kernel stay pinned to Pillow's output even where a different Pillow build is installed.

    python tests/golden/make_pil_resize_golden.py        (needs Pillow; run in the build container)
"""
import os

import numpy as np
import PIL
from PIL import Image

CASES = [  # (in_h, in_w, out_h, out_w): down + up (the JRDB stitched -> network aspect), pure down, pure up, one axis only, tiny
    (48, 376, 72, 128), (61, 97, 23, 31), (17, 19, 40, 77), (30, 50, 30, 21), (30, 50, 45, 50), (1, 1, 3, 2), (5, 300, 5, 7),
]


def main():
    out = {"pillow_version": np.array(PIL.__version__)}
    rng = np.random.default_rng(20240607)
    for k, (ih, iw, oh, ow) in enumerate(CASES):
        img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        if k % 2:                                                  # smooth content as well as noise
            yy, xx = np.mgrid[0:ih, 0:iw]
            img = np.stack([(xx * 255 // max(iw - 1, 1)), (yy * 255 // max(ih - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
        res = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        out["in_%d" % k] = img
        out["out_%d" % k] = res
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pil_resize.npz"), **out)
    print("wrote", len(CASES), "cases with Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
