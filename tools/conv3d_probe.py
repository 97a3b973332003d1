"""Times csrc/conv3d_wino.hip against the library convolution on the 3x3x3 shapes of the frozen I3D at config c3
(8 clips per pass), and prints the maximum difference between the two.

    python tools/conv3d_probe.py [clips=8] > gpurun_out/conv3d_probe.txt
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd import _lib as L  # noqa: E402

SHAPES = [  # cin, cout, d, h, w
    (64, 192, 8, 180, 320),
    (96, 128, 8, 90, 160), (16, 32, 8, 90, 160), (128, 192, 8, 90, 160), (32, 96, 8, 90, 160),
    (96, 208, 4, 45, 80), (16, 48, 4, 45, 80), (112, 224, 4, 45, 80), (24, 64, 4, 45, 80), (128, 256, 4, 45, 80),
    (144, 288, 4, 45, 80), (32, 64, 4, 45, 80), (160, 320, 4, 45, 80), (32, 128, 4, 45, 80),
]


def timed(fn, reps):
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
    clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    tot_k = tot_l = 0.0
    print("%-28s %9s %9s %7s %8s %10s" % ("shape", "kernel ms", "lib ms", "speedup", "MFMA TF", "max diff"))
    for cin, cout, d, h, w in SHAPES:
        x = torch.relu(torch.randn(clips, cin, d, h, w, device=dev))
        wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * (2.0 / (27 * cin)) ** 0.5
        y = torch.empty((clips, cout, d, h, w), device=dev)
        wp = torch.empty((L.raw("mgar_conv3d_k3_workspace_floats", cin, cout),), device=dev)

        def kern():
            L.call("mgar_conv3d_k3_fwd", L.fptr(x), clips, cin, d, h, w, L.fptr(wt), cout, L.fptr(wp), L.fptr(y), L.stream_of(x))

        def lib():
            return F.conv3d(x, wt, None, 1, 1)
        reps = 5 if cin * cout > 8000 else 20
        tk, tl = timed(kern, reps), timed(lib, reps)
        diff = (y - lib()).abs().max().item()
        issued = 2.0 * clips * d * h * w * cout * cin * 18.0
        print("%-28s %9.3f %9.3f %7.2f %8.1f %10.2e" % ("%dx%dx%dx%dx%d->%d" % (clips, cin, d, h, w, cout), tk, tl, tl / tk,
                                                       issued / tk / 1e9, diff), flush=True)
        tot_k += tk
        tot_l += tl
        del x, y, wt, wp
    print("total: kernel %.2f ms, library %.2f ms" % (tot_k, tot_l))


if __name__ == "__main__":
    main()
