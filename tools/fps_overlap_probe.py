"""When does the level-1 FPS (one 1024-thread workgroup per cloud) run if the I3D convolution is already streaming its workgroups
through the CUs -- and with a few microseconds of head start (mgar_delay_us on the convolution's stream)?
    python tools/fps_overlap_probe.py [clouds=120] [fps stream priority=0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd import _lib as L  # noqa: E402
from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as shim  # noqa: E402
from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as pb  # noqa: E402

go = []


def main():
    clouds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    dev = torch.device("cuda", 0)
    pts = torch.rand(clouds, 16384, 3, device=dev)
    clips, cin, cout, d, h, w = 8, 64, 192, 8, 180, 320
    x = torch.relu(torch.randn(clips, cin, d, h, w, device=dev))
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02
    y = torch.empty((clips, cout, d, h, w), device=dev)
    wp = torch.empty((L.raw("mgar_conv3d_k3_workspace_floats", cin, cout),), device=dev)
    prio = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)
    print('priority range', torch.cuda.Stream.priority_range(), 'FPS stream priority', prio)

    def conv():
        L.call("mgar_conv3d_k3_fwd", L.fptr(x), clips, cin, d, h, w, L.fptr(wt), cout, L.fptr(wp), L.fptr(y), sa.cuda_stream)

    def fps_ms(mode):
        """mode: 'alone', 'conv first' (convolution launched before the sampling), 'head start' (convolution 20 us after it)"""
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            if mode == "conv first":
                conv()
                conv()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(sb):
                e0.record(sb)
                if mode == "head start":
                    shim.BEFORE_SAMPLING_LAUNCH = lambda: go.append(torch.cuda.current_stream().record_event())
                pb.farthest_point_sample(pts, 4096)
                shim.BEFORE_SAMPLING_LAUNCH = None
                e1.record(sb)
            if mode == "head start":
                sa.wait_event(go.pop())
                L.call("mgar_delay_us", 20, sa.cuda_stream)
                conv()
                conv()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    pb.farthest_point_sample(pts, 4096)
    conv()
    torch.cuda.synchronize()
    print("%d clouds: FPS (Morton sort + sampling) alone %.2f ms | convolution launched first %.2f ms | convolution 20 us after the "
          "sampling kernel %.2f ms" % (clouds, fps_ms("alone"), fps_ms("conv first"), fps_ms("head start")), flush=True)


if __name__ == "__main__":
    main()
