"""One I3D 3x3x3 shape through csrc/conv3d_wino.hip, a few launches: the command profiled with rocprofv3 --pmc
(profiles/r03_conv3d_pmc.txt).    python3 tools/conv3d_one.py [cin cout d h w clips reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd import _lib as L  # noqa: E402

cin, cout, d, h, w, clips, reps = ([int(v) for v in sys.argv[1:8]] + [64, 192, 8, 180, 320, 8, 3][len(sys.argv) - 1:])[:7]
dev = torch.device("cuda", 0)
x = torch.relu(torch.randn(clips, cin, d, h, w, device=dev))
wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02
y = torch.empty((clips, cout, d, h, w), device=dev)
wp = torch.empty((L.raw("mgar_conv3d_k3_workspace_floats", cin, cout),), device=dev)
for _ in range(reps):
    L.call("mgar_conv3d_k3_fwd", L.fptr(x), clips, cin, d, h, w, L.fptr(wt), cout, L.fptr(wp), L.fptr(y), L.stream_of(x))
torch.cuda.synchronize()
print("done", float(y[0, 0, 0, 0, 0]))
