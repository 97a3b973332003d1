#!/usr/bin/env python3
"""bench.py -- clips/sec of one MGAR-net training step (forward + backward + Adam) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched by
``torch.distributed.run`` with one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

Workload = BASELINE.json config c3: 8 clips x 15 frames x 32 actors x 16 384 points, 720x1280
frames, fp32, synthetic data (multimodal_gar_amd/workload.py).  The 8 clips are the GLOBAL batch
(config c4: "same config under DDP"), sharded over the ranks => strong scaling.

Besides the headline number the line carries
  roofline     : the dominant hand-written kernel of the step, timed live with events on the stream
                 it is launched on, against the bound DESIGN.md derives for it;
  cpu_baseline : the same model code on the host cores with the CPU oracle in place of the HIP
                 kernels (oracle/cpu_backend.py), on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=8, help="GLOBAL clip batch (config c3/c4)")
    ap.add_argument("--frames", type=int, default=15)
    ap.add_argument("--actors", type=int, default=32)
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--route", default="pointnet2", choices=["pointnet2", "voxel"])
    ap.add_argument("--no-gat", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark (slow first step)")
    ap.add_argument("--no-overlap", action="store_true", help="run the RGB and LiDAR branches on one stream")
    ap.add_argument("--phases", action="store_true", help="print a synchronised per-phase timing of one step")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------
# Live timing of the hand-written kernels at the workload's shapes (events on the current stream,
# which is the stream the C ABI launches on).
# --------------------------------------------------------------------------------------------
def time_kernel(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def kernel_rooflines(points, frames_per_rank, n_points):
    """Times FPS / ball query / three_nn (the O(M*N) scans) at level-1 shapes of this rank's batch.
    Returns a list of roofline dicts, the dominant (longest per step) first."""
    from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as pb
    xyz = points[..., :3].contiguous()
    f, n = xyz.shape[0], xyz.shape[1]
    m = n // 4
    res = []
    # FPS: algorithmic bytes 12N + 4N(temp in) + 4N(temp out) + 4M per cloud; bound by the serial
    # VALU chain, so also report pair evaluations / s (10 VALU per pair)
    t = time_kernel(lambda: pb.farthest_point_sample(xyz, m), iters=3)
    by = f * (12 * n + 8 * n + 4 * m)
    pairs = f * (m - 1) * n
    res.append({"kernel": "fps_kernel<1024,%d> (N=%d -> M=%d, %d clouds)" % (max(n // 1024, 1), n, m, f), "ms": t,
                "bound": "hbm", "achieved": by / t / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": by / t / 1e6 / HBM_PEAK_GBS, "traffic": None,
                "pair_evals_per_s": pairs / (t * 1e-3), "valu_frac": pairs * 10 * 2 / (t * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                "per_step_launches": 1})
    idx = pb.farthest_point_sample(xyz, m)
    new_xyz = torch.gather(xyz, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
    for radius, ns in ((0.1, 16), (0.5, 32)):
        t = time_kernel(lambda: pb.ball_query(radius, ns, xyz, new_xyz))
        by = f * (12 * n + 12 * m + 4 * m * ns)
        pairs = f * m * n
        res.append({"kernel": "ball_query_kernel<batch> (r=%.1f, ns=%d, M=%d x N=%d, %d clouds)" % (radius, ns, m, n, f),
                    "ms": t, "bound": "hbm", "achieved": by / t / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": by / t / 1e6 / HBM_PEAK_GBS, "traffic": None, "pair_evals_per_s": pairs / (t * 1e-3),
                    "valu_frac": pairs * 7 * 2 / (t * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, "per_step_launches": 1})
    t = time_kernel(lambda: pb.three_nn(xyz, new_xyz))
    by = f * (12 * n + 12 * m + 24 * n)
    res.append({"kernel": "three_nn_kernel<batch> (n=%d x m=%d, %d clouds)" % (n, m, f), "ms": t, "bound": "hbm",
                "achieved": by / t / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / t / 1e6 / HBM_PEAK_GBS,
                "traffic": None, "pair_evals_per_s": f * n * m / (t * 1e-3),
                "valu_frac": f * n * m * 8 * 2 / (t * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, "per_step_launches": 1})
    res.sort(key=lambda r: -r["ms"])
    return res


# --------------------------------------------------------------------------------------------
# CPU baseline: same Python model code, oracle kernels, host cores, bounded sample.
# --------------------------------------------------------------------------------------------
def cpu_baseline(args):
    from multimodal_gar_amd import workload as W
    from oracle import oracle as O
    from oracle.cpu_backend import use_cpu_oracle
    # the GPU box hands one GPU a share of the host: use the CPUs this process may run on,
    # capped at 16 (oversubscribing the 256 logical CPUs the OS reports is 10x slower)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = args.cpu_threads or max(1, min(avail, 16))
    torch.set_num_threads(cores)
    O.build(); O.set_threads(cores)
    dev = torch.device("cpu")
    sample_frames = min(3, args.frames)
    batch = W.make_batch(7, 1, sample_frames, args.actors, args.points, args.height, args.width, dev)
    with use_cpu_oracle():
        step = W.TrainStep(args.actors, args.points, dev, gat=not args.no_gat, route=args.route)
        step.run(batch)                                     # warm-up (allocator, oneDNN primitive cache)
        t0 = time.time(); step.module.rgb_tokens(batch["images"], batch["bboxes"]); t_rgb = time.time() - t0
        t0 = time.time(); step.run(batch); t_all = time.time() - t0
    t_frames = max(t_all - t_rgb, 1e-6)
    scale = args.frames / sample_frames
    clip_s = (t_rgb + t_frames) * scale
    return {"value": 1.0 / clip_s, "unit": "clips/sec", "cores": cores, "kind": "port",
            "sample": "1 clip x %d frames (of %d) at full A=%d, P=%d, %dx%d, fwd+bwd+Adam, after one warm-up pass; "
                      "time scaled x%.0f to a %d-frame clip; %.1f s measured (I3D part %.1f s)"
                      % (sample_frames, args.frames, args.actors, args.points, args.height, args.width, scale,
                         args.frames, t_all, t_rgb)}


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def phase_timing(step, batch):
    """One forward/backward with a device sync after each phase (diagnostic; not the timed region)."""
    import torch
    m = step.module
    out = {}

    def tick(name, t0):
        torch.cuda.synchronize(); out[name] = (time.perf_counter() - t0) * 1e3; log("phase %-12s %9.1f ms" % (name, out[name]))
    step.opt.zero_grad(set_to_none=True)
    t0 = time.perf_counter(); rgb = m.rgb_tokens(batch["images"], batch["bboxes"]); tick("rgb_fwd", t0)
    if m.route == "pointnet2":   # split the LiDAR forward: SA/FP trunk, RoI-grid lift, NL block + embedding
        import torch as _t
        pts, b3 = batch["points"], batch["bboxes3d"]
        f, p, _ = pts.shape
        lb = m.net.LiDAR_backbone
        with _t.no_grad():
            bidx = _t.arange(f, device=pts.device, dtype=pts.dtype).view(f, 1, 1).expand(f, p, 1)
            data = {"batch_size": f, "points": _t.cat([bidx, pts], -1).view(f * p, 5), "gt_boxes": b3[:, :m.n_actors, :].contiguous(),
                    "point_batch_cnt": _t.full((f,), p, dtype=_t.int32, device=pts.device)}
            t0 = time.perf_counter(); data = lb.model.backbone_3d(data); tick(" sa_fp_trunk", t0)
            t0 = time.perf_counter(); data = lb.model.roi_head(data); tick(" roi_grid_lift", t0)
    t0 = time.perf_counter(); lidar = m.lidar_tokens(batch["points"], batch["bboxes3d"]); tick("lidar_fwd", t0)
    b, t, a = batch["n_clips"], batch["n_frames"], m.n_actors
    t0 = time.perf_counter()
    rgb_s = rgb[:, None].expand(b, t, a, rgb.shape[-1]).reshape(b * t, a, -1)
    pad = lambda x: torch.cat([x, x.new_zeros(x.shape[0], 1, x.shape[2])], 1)  # noqa: E731
    bb2 = batch["bboxes"][:, None].expand(b, t, a + 1, 4).reshape(b * t, a + 1, 4)
    res = m.net.GAR_model(pad(rgb_s), pad(lidar), bb2, batch["bboxes3d"], None, batch["person_id"])
    from multimodal_gar_amd.workload import synthetic_loss
    loss = synthetic_loss(res); tick("fusion_fwd", t0)
    t0 = time.perf_counter(); loss.backward(); tick("backward", t0)
    t0 = time.perf_counter(); step.opt.step(); tick("adam", t0)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels have no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ddp = world > 1
    if ddp:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)
    assert args.clips % world == 0, "global clip batch must divide over the ranks"
    clips_local = args.clips // world

    from multimodal_gar_amd import workload as W
    torch.backends.cudnn.benchmark = bool(args.miopen_find)   # MIOpen find mode for the I3D convolutions
    log("building model (rank %d/%d, %d clips on this rank)" % (rank, world, clips_local))
    step = W.TrainStep(args.actors, args.points, dev, gat=not args.no_gat, route=args.route, ddp=ddp)
    step.module.overlap_branches = not args.no_overlap
    batch = W.make_batch(100 + rank, clips_local, args.frames, args.actors, args.points, args.height, args.width, dev)

    def barrier():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    log("model + batch ready; %.1f GB allocated" % (torch.cuda.memory_allocated() / 2 ** 30))
    if args.phases and rank == 0 and not ddp:
        phase_timing(step, batch)
        phase_timing(step, batch)
    for i in range(args.warmup):
        step.run(batch); torch.cuda.synchronize(); log("warmup step %d done" % i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step.run(batch)
        if rank == 0:
            log("timed step %d issued" % i)   # host-side only: no sync inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    log("timed region: %.1f ms/step, peak mem %.1f GB" % (elapsed / args.steps * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30))
    if ddp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = args.clips * args.steps / elapsed

    roof, kernels, cpu = None, None, None
    if rank == 0 and not args.no_kernel_timing:
        kernels = kernel_rooflines(batch["points"], clips_local * args.frames, args.points)
        roof = dict(kernels[0])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    if rank == 0:
        line = {
            "metric": "clips/sec (fwd+bwd) at 32 actors x 16k pts x 15 frames",
            "value": value, "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c3: %d clips x %d frames x %d actors x %d pts, %dx%d RGB, fp32 fwd+bwd+Adam, "
                                   "LiDAR route %s, GAT %s" % (args.clips, args.frames, args.actors, args.points,
                                                                args.height, args.width, args.route,
                                                                "off" if args.no_gat else "on"),
                       "global_clips": args.clips, "clips_per_gpu": clips_local, "parallelism": "dp%d" % world,
                       "trainable_params": W.trainable_parameter_count(step.module)},
            "roofline": roof, "cpu_baseline": cpu, "kernels": kernels,
        }
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
