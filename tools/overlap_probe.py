"""Does an HBM-bound stream run BESIDE the MFMA-bound I3D convolution, or only between its workgroups?

Stream A: csrc/conv3d_wino.hip on Conv3d_2c_3x3's shape (8 clips).  Stream B: a chain of the LiDAR forward's streaming kernels
(pointwise_conv_fwd with BatchNorm + ReLU prologue, a device copy).  Prints A alone, B alone, A || B, for the convolution at
two workgroups per CU (default) and at one (extra dynamic LDS).

    python tools/overlap_probe.py > gpurun_out/overlap_probe.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd import _lib as L  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    clips, cin, cout, d, h, w = 8, 64, 192, 8, 180, 320
    x = torch.relu(torch.randn(clips, cin, d, h, w, device=dev))
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02
    y = torch.empty((clips, cout, d, h, w), device=dev)
    wp = torch.empty((L.raw("mgar_conv3d_k3_workspace_floats", cin, cout),), device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    f, ci, co, p = 120, 32, 32, 131072
    px = torch.randn(f, ci, p, device=dev)
    pw = torch.randn(co, ci, device=dev)
    py = torch.empty(f, co, p, device=dev)
    mean = torch.zeros(ci, device=dev)
    invstd = torch.ones(ci, device=dev)
    big = torch.empty(512 * 1024 * 1024, device=dev)      # 2 GB copy source
    big2 = torch.empty_like(big)

    def conv():
        with torch.cuda.stream(sa):
            L.call("mgar_conv3d_k3_fwd", L.fptr(x), clips, cin, d, h, w, L.fptr(wt), cout, L.fptr(wp), L.fptr(y), sa.cuda_stream)

    def stream_b(kind, reps):
        with torch.cuda.stream(sb):
            for _ in range(reps):
                if kind == "pointwise":
                    L.call("mgar_pointwise_conv_fwd", L.fptr(px), f, ci, p, L.fptr(pw), ci, 1, co, L.fptr(mean), L.fptr(invstd), None, None, 1,
                           L.fptr(py), sb.cuda_stream)
                else:
                    big2.copy_(big)

    def wall(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(torch.cuda.current_stream())
        sa.wait_event(e0); sb.wait_event(e0)
        fn()
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(sa); eb.record(sb)
        torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    for pad in (0, 40 * 1024):
        L.call("mgar_conv3d_k3_set_lds_pad", pad)
        ta = wall(conv)
        print("conv3d 64->192 alone, lds pad %d KB: %.2f ms" % (pad // 1024, ta), flush=True)
        for kind, reps in (("pointwise", 16), ("copy", 12)):
            tb = wall(lambda: stream_b(kind, reps))
            tab = wall(lambda: (conv(), stream_b(kind, reps)))
            tba = wall(lambda: (stream_b(kind, reps), conv()))
            print("   B = %2d x %-9s alone %.2f ms | conv first || B: %.2f ms | B first || conv: %.2f ms | sum %.2f ms" %
                  (reps, kind, tb, tab, tba, ta + tb), flush=True)
    L.call("mgar_conv3d_k3_set_lds_pad", 0)


if __name__ == "__main__":
    main()
