"""Probe: does MIOpen run conv3d on channels_last_3d tensors without its batched_transpose adapters, and how fast?"""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_gar_amd  # noqa: F401  (MIOpen user db set-up)
torch.backends.cudnn.benchmark = True
dev = "cuda"
for (n, cin, t, h, w, cout, k) in [(8, 64, 8, 180, 320, 192, 3), (8, 128, 8, 90, 160, 192, 3), (8, 192, 8, 90, 160, 64, 1)]:
    x = torch.randn(n, cin, t, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, k, device=dev) * 0.05
    for fmt in ("contiguous", "channels_last_3d"):
        xx = x.contiguous(memory_format=torch.channels_last_3d) if fmt != "contiguous" else x
        ww = wt.contiguous(memory_format=torch.channels_last_3d) if fmt != "contiguous" else wt
        for _ in range(3):
            y = F.conv3d(xx, ww, padding=k // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            y = F.conv3d(xx, ww, padding=k // 2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5 * 1e3
        print("%-18s %s k=%d  %.2f ms  out strides %s channels_last=%s" % (fmt, (n, cin, t, h, w, cout), k, dt, y.stride(),
              y.is_contiguous(memory_format=torch.channels_last_3d)), flush=True)
