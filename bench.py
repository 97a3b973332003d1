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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import multimodal_gar_amd  # noqa: E402,F401  -- points MIOPEN_USER_DB_PATH at a scratch copy of the shipped find-db before MIOpen starts

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak, same guide
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA, same guide
BF16_MFMA_KERNELS = {"stem_conv3d_kernel"}   # hand-written kernels whose bf16-payload variant multiplies on the bf16 MFMA


# BASELINE.json configs that run on one rank.  c3 is the metric's configuration (and c4 = c3 sharded over ranks);
# c2 / c5 are the bf16 configurations (feature payloads and GEMMs in bf16, coordinates / distances / indices fp32 / int32,
# SURVEY.md section 8 header) and are forward passes.
CONFIGS = {
    "c3": dict(clips=8, frames=15, actors=32, points=16384, precision="fp32", mode="train"),
    "c2": dict(clips=4, frames=15, actors=16, points=8192, precision="bf16", mode="forward"),
    "c5": dict(clips=8, frames=15, actors=128, points=65536, precision="bf16", mode="forward"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration: c3 = the metric's (fp32 fwd+bwd+Adam, 8 clips x 32 actors x 16 384 pts); "
                         "c2 = bf16 forward, 4 clips x 16 actors x 8 192 pts; c5 = bf16 forward, 128 actors x 65 536 pts")
    ap.add_argument("--clips", type=int, default=None, help="GLOBAL clip batch (default: the configuration's)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--actors", type=int, default=None)
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16"])
    ap.add_argument("--mode", default=None, choices=["train", "forward"],
                    help="train = forward + backward + Adam; forward = train-mode forward only (batch-statistics BatchNorm)")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--route", default="pointnet2", choices=["pointnet2", "voxel"])
    ap.add_argument("--no-gat", action="store_true")
    ap.add_argument("--unfused-voxel-pool", action="store_true",
                    help="--route voxel only: run the reference-shaped op chain of NeighborVoxelSAModuleMSG on the device instead "
                         "of the fused csrc/voxel_roi_pool.hip kernels (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-miopen-find", action="store_true",
                    help="do not use MIOpen find mode (torch.backends.cudnn.benchmark) for the I3D convolutions; with it the "
                         "solver choice comes from multimodal_gar_amd/miopen_db (24 %% faster I3D than immediate mode)")
    ap.add_argument("--no-overlap", action="store_true", help="run the RGB and LiDAR branches on one stream")
    ap.add_argument("--i3d-channels-last", action="store_true",
                    help="keep the I3D activations NDHWC (no MIOpen layout adapters: ~3 ms/step faster at c3, but MIOpen's kernel "
                         "search adds ~4 minutes to the start-up of a fresh process)")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue every kernel from the host instead of replaying forward + backward from a HIP graph "
                         "(torch.cuda.CUDAGraph; default on: 4-9 %% per step, more on ranks that hold a single clip)")
    ap.add_argument("--ddp-wrapper", action="store_true",
                    help="eager DistributedDataParallel (bucketed all-reduce overlapped with backward) instead of one flattened "
                         "gradient all-reduce after the backward; implies --no-graph (DDP hooks cannot be captured)")
    ap.add_argument("--phases", action="store_true", help="print a synchronised per-phase timing of one step")
    args = ap.parse_args()
    for k, v in CONFIGS[args.config].items():
        if getattr(args, k) is None:
            setattr(args, k, v)
    return args


# --------------------------------------------------------------------------------------------
# Live timing of the hand-written kernels: the library brackets each instrumented kernel launch with
# two HIP events recorded on the stream the kernel is launched on (csrc/errors.hip, mgar_ktimer_*) and
# notes the launch's algorithmic bytes / flops.  The timers
# are switched on for ONE extra step of the same workload right after the timed region -- issued eagerly on a
# single stream, so that every kernel has the chip to itself between its two events -- and the headline number
# is not perturbed by ~3 000 event records per step.
# --------------------------------------------------------------------------------------------
MFMA_KERNELS = ("pointwise_fwd_kernel", "pointwise_dw_kernel", "rowmajor_dw_kernel", "stem_conv3d_kernel")
PAIR_KERNELS = ("fps_kernel", "ball_query_kernel", "three_nn_kernel")   # (query, point) scans: VALU-bound, pair evaluations of 8 flop
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_hbm_traffic.json")   # rocprofv3 --pmc passes, see profiles/README.md


def pmc_traffic(clips_local):
    """-> ({kernel: bytes per launch}, source tag).  The PMC figures are a committed measurement (tools/pmc_traffic.py on
    two rocprofv3 --pmc passes), NOT taken in this run: a figure is attached only for the per-rank batch it was measured
    at and only while the kernel's .hip source still has the sha it had then."""
    from multimodal_gar_amd.op_timer import source_sha16
    if not os.path.exists(PMC_TRAFFIC_FILE):
        return {}, None
    try:
        pmc = json.load(open(PMC_TRAFFIC_FILE))
    except (OSError, ValueError):
        return {}, None
    if pmc.get("clips_per_gpu") != clips_local:
        return {}, None
    shas = pmc.get("source_sha16", {})
    ok = {k: v for k, v in pmc.get("per_launch_bytes", {}).items() if shas.get(k) and shas.get(k) == source_sha16(k)}
    tag = "profiles/pmc_hbm_traffic.json@%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not measured in this run)" \
        % pmc.get("commit", "unknown")
    return ok, tag


def kernel_rooflines(step, batch, frames, n_points, clips_local=None, precision="fp32"):
    """One extra, instrumented step (eager, ONE stream) -> (rows, accounting).  rows: one roofline dict per hand-written
    kernel (events inside the library, csrc/errors.hip) AND per library op / shape (convolutions and GEMMs with their
    FLOPs against the MFMA peak of their dtype; multimodal_gar_amd/op_timer.py), the largest total time first.
    accounting: where the whole step goes, by class."""
    from multimodal_gar_amd import _lib as L
    from multimodal_gar_amd.op_timer import AtenOpTimer
    # one stream for this step: with the RGB branch on its side stream two kernels share the chip and the time
    # between a kernel's two events is no longer the time that kernel needs
    overlap, step.module.overlap_branches = step.module.overlap_branches, False
    # one plain eager step first: the timed steps replayed a graph (private memory pool) and capture()'s warm-up ran on a
    # side stream, so this stream's caching-allocator pool is cold -- without this every large allocation of the
    # instrumented step would be a synchronous hipMalloc sitting between an op's two events
    step.run_eager(batch)
    torch.cuda.synchronize()
    L.kernel_timers(enable=True)
    L.kernel_timers()                       # drop anything recorded so far
    t0 = time.perf_counter()
    with AtenOpTimer() as lib_ops:
        step.run_eager(batch)
        torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) * 1e3
    L.kernel_timers(enable=False)
    step.module.overlap_branches = overlap
    table = L.kernel_timers()
    traffic, traffic_src = pmc_traffic(clips_local)
    res = []
    for name, (ms, launches, nbytes, flops) in table.items():
        gbs = nbytes / ms / 1e6 if ms > 0 else 0.0
        row = {"kernel": name, "class": "hand_written", "launches_per_step": launches, "ms_per_step": ms, "avg_launch_ms": ms / launches,
               "algorithmic_bytes_per_launch": nbytes / launches, "traffic": traffic.get(name),
               "traffic_source": traffic_src if name in traffic else None}
        if name in MFMA_KERNELS:
            tf = flops / ms / 1e9 if ms > 0 else 0.0
            # the I3D stem runs on the bf16 MFMA (2.5 PF dense) when the payloads are bf16; everything else here is exact fp32 MFMA
            peak = BF16_MFMA_PEAK_TFLOPS if (precision == "bf16" and name in BF16_MFMA_KERNELS) else VALU_PEAK_TFLOPS
            row.update({"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                        "mfma_dtype": "bf16" if peak == BF16_MFMA_PEAK_TFLOPS else "f32", "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
        else:
            row.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS})
        if name in PAIR_KERNELS and flops:
            # these scans move almost nothing (the HBM fraction above is the contract's figure and says so); what bounds
            # them is fp32 VALU work: pair evaluations per second and their share of the vector peak
            row["pair_evals_per_s"] = flops / 8.0 / (ms * 1e-3)
            row["valu_frac"] = flops / (ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS
        res.append(row)
    lib_rows, lib_totals = lib_ops.table(top=int(os.environ.get("MGAR_BENCH_LIB_ROWS", "12")))   # rows per class (conv / gemm / other)
    res += lib_rows
    res.sort(key=lambda r: -r["ms_per_step"])
    for r in res:
        if r.get("frac", 0.0) > 1.0:
            r["suspect"] = "fraction above the peak: the algorithmic figure over-counts what this kernel must move"
            log("WARNING: %s reports %.2f of its roofline" % (r["kernel"], r["frac"]))
    hand = sum(ms for ms, _, _, _ in table.values())
    accounting = {"hand_written_kernels_ms": hand, "library_conv_ms": lib_totals["conv"], "library_gemm_ms": lib_totals["gemm"],
                  "torch_elementwise_copy_reduce_ms": lib_totals["other"],
                  "sum_ms": hand + sum(lib_totals.values()), "instrumented_step_wall_ms": step_ms,
                  "note": "one eager single-stream step with events around every kernel / aten op (slower than the timed "
                          "graph-replayed two-stream steps; its purpose is attribution)"}
    return res, accounting


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
    passes = 3
    with use_cpu_oracle():
        step = W.TrainStep(args.actors, args.points, dev, gat=not args.no_gat, route=args.route)
        step.run(batch)                                     # warm-up (allocator, oneDNN primitive cache)
        t_rgbs, t_alls = [], []
        for _ in range(passes):
            t0 = time.time(); step.module.rgb_tokens(batch["images"], batch["bboxes"]); t_rgbs.append(time.time() - t0)
            t0 = time.time(); step.run(batch); t_alls.append(time.time() - t0)
    t_rgb, t_all = sorted(t_rgbs)[passes // 2], sorted(t_alls)[passes // 2]       # medians
    t_frames = max(t_all - t_rgb, 1e-6)
    scale = args.frames / sample_frames
    clip_s = (t_rgb + t_frames) * scale          # the sample clip has sample_frames RGB frames AND LiDAR frames: both parts scale
    return {"value": 1.0 / clip_s, "unit": "clips/sec", "cores": cores, "kind": "port", "timed_passes": passes,
            "sample": "1 clip x %d frames (of %d) at full A=%d, P=%d, %dx%d, fwd+bwd+Adam: 1 warm-up + %d timed passes, median; "
                      "time scaled x%.0f to a %d-frame clip; "
                      "step %.1f s (min %.1f, max %.1f), I3D part %.1f s"
                      % (sample_frames, args.frames, args.actors, args.points, args.height, args.width, passes, scale,
                         args.frames, t_all, min(t_alls), max(t_alls), t_rgb)}


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
    loss = step._loss_of(res, batch); tick("fusion_fwd", t0)
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
    # MGAR_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a one-GPU box; RCCL
    # refuses two ranks on one device); the real runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("MGAR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ddp = world > 1
    if ddp:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert args.clips % world == 0, "global clip batch must divide over the ranks"
    clips_local = args.clips // world

    from multimodal_gar_amd import workload as W
    torch.backends.cudnn.benchmark = not args.no_miopen_find   # MIOpen find mode for the I3D convolutions
    log("building model (rank %d/%d, %d clips on this rank)" % (rank, world, clips_local))
    use_graph = not args.no_graph and not args.ddp_wrapper
    if use_graph and args.route == "voxel":
        # the voxeliser in front of the voxel route (torch.unique, data-dependent sizes) synchronises the host and cannot be
        # captured: this route is timed with host-issued launches on one stream
        log("route voxel: host-issued launches (the voxeliser's data-dependent sizes cannot be captured into a graph)")
        use_graph = False
        args.no_graph = True
    if args.mode == "train":
        if args.precision != "fp32":
            raise SystemExit("bench.py: the backward runs in fp32 only (bf16 is a forward configuration: c2 / c5)")
        step = W.TrainStep(args.actors, args.points, dev, gat=not args.no_gat, route=args.route, ddp=ddp,
                           manual_allreduce=not args.ddp_wrapper)
    else:
        step = W.ForwardStep(args.actors, args.points, dev, gat=not args.no_gat, route=args.route, precision=args.precision)
    if args.unfused_voxel_pool:
        from multimodal_gar_amd.pcdet.ops.pointnet2.pointnet2_stack.voxel_pool_modules import NeighborVoxelSAModuleMSG
        for m in step.module.modules():
            if isinstance(m, NeighborVoxelSAModuleMSG):
                m.fused = False
    # frozen I3D on a side stream (no autograd there: DDP-safe).  Only together with the HIP graph: issued eagerly from
    # the host the two-stream step measured 362 ms against 269 ms on one stream (and 257 ms as a graph on two).
    step.module.overlap_branches = not args.no_overlap and not args.no_graph and not args.ddp_wrapper
    step.module.i3d_channels_last = bool(args.i3d_channels_last)
    batch = W.make_batch(100 + rank, clips_local, args.frames, args.actors, args.points, args.height, args.width, dev)

    def barrier():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    log("model + batch ready; %.1f GB allocated" % (torch.cuda.memory_allocated() / 2 ** 30))
    if args.phases and rank == 0 and not ddp:
        phase_timing(step, batch)
        phase_timing(step, batch)
    if use_graph:
        try:
            step.capture(batch)
            log("forward + backward captured into a HIP graph")
        except RuntimeError as e:   # what torch raises for an op that cannot be captured / a failed HIP call during capture
            # a failed capture leaves the device RNG in capture mode: nothing after it can be trusted, so this is fatal
            # (run with --no-graph to time host-issued launches)
            raise SystemExit("bench.py: HIP-graph capture failed (%s: %s); rerun with --no-graph"
                             % (type(e).__name__, str(e).splitlines()[0][:200]))
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

    roof, kernels, cpu, accounting = None, None, None, None
    if not args.no_kernel_timing:
        if rank == 0:
            kernels, accounting = kernel_rooflines(step, batch, clips_local * args.frames, args.points, clips_local, args.precision)
            # the dominant kernel of the WHOLE step (hand-written or library) that has a roofline; and, beside it, the
            # dominant hand-written one (the kernels this repo can tune)
            dom = next(r for r in kernels if "bound" in r and "frac" in r)
            own = next((r for r in kernels if r.get("class") == "hand_written"), None)
            keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "class", "launches_per_step",
                    "ms_per_step", "avg_launch_ms", "algorithmic_bytes_per_launch", "flops_per_launch", "dtype")
            roof = {k: dom[k] for k in keys if k in dom}
            roof.setdefault("traffic", None)
            if own is not None and own is not dom:
                roof["dominant_hand_written"] = {k: own[k] for k in keys if k in own}
            log("dominant kernel of the step: %s, %.2f ms/step in %d launches, %.0f %s (%.1f %% of peak)"
                % (roof["kernel"], roof["ms_per_step"], roof["launches_per_step"], roof["achieved"], roof["unit"], 100 * roof["frac"]))
            log("step accounting (ms): %s" % {k: (round(v, 1) if isinstance(v, float) else v) for k, v in accounting.items() if k != "note"})
        else:
            step.run_eager(batch)    # the extra (instrumented on rank 0) step is collective under DDP
        barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "train":
        cpu = cpu_baseline(args)
    if rank == 0:
        what = "fwd+bwd" if args.mode == "train" else "fwd"
        line = {
            "metric": "clips/sec (%s) at %d actors x %dk pts x %d frames" % (what, args.actors, args.points // 1024, args.frames),
            "value": value, "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "%s: %d clips x %d frames x %d actors x %d pts, %dx%d RGB, %s %s, "
                                   "LiDAR route %s, GAT %s" % (args.config, args.clips, args.frames, args.actors, args.points,
                                                                args.height, args.width, args.precision,
                                                                "fwd+bwd+Adam" if args.mode == "train" else
                                                                "train-mode forward (feature payloads + GEMMs bf16; xyz, distances, "
                                                                "indices, BN statistics fp32/int32)" if args.precision == "bf16"
                                                                else "train-mode forward",
                                                                args.route, "off" if args.no_gat else "on"),
                       "global_clips": args.clips, "clips_per_gpu": clips_local, "parallelism": "dp%d" % world,
                       "launch": "hip_graph" if step.graph is not None else "eager",
                       "gradient_exchange": "none" if world == 1 else ("ddp_bucketed" if args.ddp_wrapper else "flat_allreduce"),
                       "trainable_params": W.trainable_parameter_count(step.module)},
            "roofline": roof, "cpu_baseline": cpu, "step_accounting": accounting, "kernels": kernels,
        }
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
