"""Spatial partition of the chip between an MFMA-bound and an HBM-bound stream with per-stream CU masks
(hipExtStreamCreateWithCUMask): how do csrc/conv3d_wino.hip (Conv3d_2c_3x3's shape) and a chain of streaming kernels scale
with the number of CUs each gets, and what does running them side by side on disjoint CU sets cost compared with one after
the other?

    python tools/cumask_probe.py > gpurun_out/cumask_probe.txt
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd import _lib as L  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
TOTAL_CUS = 256


def masked_stream(bits):
    """bits: iterable of CU indices to enable."""
    words = (TOTAL_CUS + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for b in bits:
        mask[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, "hipExtStreamCreateWithCUMask failed: %d" % rc
    return torch.cuda.ExternalStream(s.value)


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.zeros(1, device=dev)
    clips, cin, cout, d, h, w = 8, 64, 192, 8, 180, 320
    x = torch.relu(torch.randn(clips, cin, d, h, w, device=dev))
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02
    y = torch.empty((clips, cout, d, h, w), device=dev)
    wp = torch.empty((L.raw("mgar_conv3d_k3_workspace_floats", cin, cout),), device=dev)
    f, ci, co, p = 120, 32, 32, 131072
    px = torch.randn(f, ci, p, device=dev)
    pw = torch.randn(co, ci, device=dev)
    py = torch.empty(f, co, p, device=dev)
    mean = torch.zeros(ci, device=dev)
    invstd = torch.ones(ci, device=dev)
    big = torch.empty(512 * 1024 * 1024, device=dev)
    big2 = torch.empty_like(big)

    def conv(s):
        L.call("mgar_conv3d_k3_fwd", L.fptr(x), clips, cin, d, h, w, L.fptr(wt), cout, L.fptr(wp), L.fptr(y), s.cuda_stream)

    def chain(s, kind, reps):
        with torch.cuda.stream(s):
            for _ in range(reps):
                if kind == "pointwise":
                    L.call("mgar_pointwise_conv_fwd", L.fptr(px), f, ci, p, L.fptr(pw), ci, 1, co, L.fptr(mean), L.fptr(invstd), None, None, 1,
                           L.fptr(py), s.cuda_stream)
                else:
                    big2.copy_(big)

    def wall(streams, fn, reps=3):
        best = 1e9
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            e0.record(cur)
            for s in streams:
                s.wait_event(e0)
            fn()
            for s in streams:
                ev = torch.cuda.Event()
                ev.record(s)
                cur.wait_event(ev)
            e1.record(cur)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
    print("unmasked: conv %.2f ms | 16 x pointwise %.2f ms | 12 x copy %.2f ms" % (
        wall([plain_a], lambda: conv(plain_a)), wall([plain_b], lambda: chain(plain_b, "pointwise", 16)),
        wall([plain_b], lambda: chain(plain_b, "copy", 12))), flush=True)
    print("unmasked side by side: conv || pointwise %.2f ms, conv || copy %.2f ms" % (
        wall([plain_a, plain_b], lambda: (conv(plain_a), chain(plain_b, "pointwise", 16))),
        wall([plain_a, plain_b], lambda: (conv(plain_a), chain(plain_b, "copy", 12)))), flush=True)
    for n_a in (224, 192, 160, 128):
        sa = masked_stream(range(0, n_a))
        sb = masked_stream(range(n_a, TOTAL_CUS))
        ta = wall([sa], lambda: conv(sa))
        tp = wall([sb], lambda: chain(sb, "pointwise", 16))
        tc = wall([sb], lambda: chain(sb, "copy", 12))
        tap = wall([sa, sb], lambda: (conv(sa), chain(sb, "pointwise", 16)))
        tac = wall([sa, sb], lambda: (conv(sa), chain(sb, "copy", 12)))
        print("conv on %3d CUs: %.2f ms | on the other %3d: pointwise %.2f ms, copy %.2f ms | side by side: conv || pointwise %.2f ms, "
              "conv || copy %.2f ms" % (n_a, ta, TOTAL_CUS - n_a, tp, tc, tap, tac), flush=True)


if __name__ == "__main__":
    main()
