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
torch = None   # imported by _import_torch(): `bench.py --gpus N` must be able to start its ranks before torch / the package load


def _import_torch():
    global torch
    import multimodal_gar_amd  # noqa: F401  -- points MIOPEN_USER_DB_PATH at a scratch copy of the shipped find-db before MIOpen starts
    import torch as _torch
    torch = _torch
    return torch


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
    ap.add_argument("--prefetch-rgb", action="store_true",
                    help="software-pipeline the FROZEN RGB branch: every step runs I3D + RoIAlign for the NEXT batch under its own "
                         "backward instead of beside the LiDAR forward (same work per step; off by default)")
    ap.add_argument("--prefetch-geometry", action="store_true",
                    help="input-side software pipelining (off by default): every step computes the PointNet++ trunk's coordinate-only "
                         "work (FPS, ball queries, 3-NN weights) for the NEXT batch on a side stream while it runs the feature path and "
                         "the backward of the current one; the same work per step, the level-1 FPS off the critical path")
    ap.add_argument("--phases", action="store_true", help="print a synchronised per-phase timing of one step")
    ap.add_argument("--kernels-out", default=None,
                    help="where the per-kernel roofline table of the instrumented step is written (default: "
                         "gpurun_out/bench_kernels_<config>_<route>_<N>gpu_<clips>clips.json; the stdout line only names it)")
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
MFMA_KERNELS = ("pointwise_fwd_kernel", "pointwise_dw_kernel", "rowmajor_dw_kernel", "stem_conv3d_kernel", "spconv_gemm", "spconv_dw", "conv3d_wino_kernel")
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


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def extra_steps(step, batch, instrument):
    """The two extra eager steps every rank issues after the timed region -- the SAME number of steps (hence of gradient
    exchanges) on every rank; only what surrounds the second one differs: `instrument` (rank 0) brackets it with the
    library's kernel timers and the aten-op timer.  -> (kernel table, AtenOpTimer, wall ms) or None.

    One stream for these steps: with the RGB branch on its side stream two kernels share the chip and the time between a
    kernel's two events is no longer the time that kernel needs.  The first, plain step is there because the timed steps
    replayed a graph (private memory pool) and capture()'s warm-up ran on a side stream, so this stream's caching-allocator
    pool is cold -- without it every large allocation of the instrumented step would be a synchronous hipMalloc sitting
    between an op's two events."""
    overlap, step.module.overlap_branches = step.module.overlap_branches, False
    try:
        step.run_eager(batch)
        _sync()
        if not instrument:
            step.run_eager(batch)
            _sync()
            return None
        from multimodal_gar_amd import _lib as L
        from multimodal_gar_amd.op_timer import AtenOpTimer
        L.kernel_timers(enable=True)
        L.kernel_timers()                       # drop anything recorded so far
        t0 = time.perf_counter()
        try:
            with AtenOpTimer() as lib_ops:
                step.run_eager(batch)
                _sync()
        finally:
            L.kernel_timers(enable=False)
        step_ms = (time.perf_counter() - t0) * 1e3
        return L.kernel_timers(), lib_ops, step_ms
    finally:
        step.module.overlap_branches = overlap


def kernel_rooflines(table, lib_ops, step_ms, clips_local=None, precision="fp32"):
    """The instrumented step's records -> (rows, accounting).  rows: one roofline dict per hand-written kernel (events
    inside the library, csrc/errors.hip) AND per library op / shape (convolutions and GEMMs with their FLOPs against the
    MFMA peak of their dtype; multimodal_gar_amd/op_timer.py), the largest total time first.  accounting: where the whole
    step goes, by class."""
    traffic, traffic_src = pmc_traffic(clips_local)
    res = []
    for name, (ms, launches, nbytes, flops) in table.items():
        gbs = nbytes / ms / 1e6 if ms > 0 else 0.0
        row = {"kernel": name, "class": "hand_written", "launches_per_step": launches, "ms_per_step": ms, "avg_launch_ms": ms / launches,
               "algorithmic_bytes_per_launch": nbytes / launches, "traffic": traffic.get(name),
               "traffic_source": traffic_src if name in traffic else None}
        if flops:
            row["flops_per_launch"] = flops / launches
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
                  "sum_ms": hand + sum(lib_totals.values()), "instrumented_step_wall_ms": step_ms}
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


LINE_LIMIT = 4096   # bytes: the driver keeps a bounded tail of stdout (BENCH_r02: a 22 KB line lost its head and parsed as null)
ROOF_KEYS = ("kernel", "class", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "ms_per_step",
             "avg_launch_ms", "algorithmic_bytes_per_launch", "flops_per_launch", "dtype", "mfma_dtype", "valu_frac")


def _round(x, digits=5):
    """Floats to `digits` significant digits, recursively (line size)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _round(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, digits) for v in x]
    return x


def roofline_entry(kernels):
    """The `roofline` object of the line: the dominant kernel of the WHOLE step (hand-written or library) that has a
    roofline, and beside it the dominant hand-written one (the kernels this repo can tune)."""
    dom = next(r for r in kernels if "bound" in r and "frac" in r)
    own = next((r for r in kernels if r.get("class") == "hand_written"), None)
    roof = {k: dom[k] for k in ROOF_KEYS if k in dom}
    roof.setdefault("traffic", None)
    if own is not None and own is not dom:
        roof["dominant_hand_written"] = {k: own[k] for k in ROOF_KEYS if k in own}
        roof["dominant_hand_written"].setdefault("traffic", None)
    roof["measured_in"] = ("one eager single-stream step issued after the timed region, HIP events around every kernel / "
                           "aten op (the timed steps replay a two-stream HIP graph); traffic = committed rocprofv3 --pmc "
                           "passes (profiles/pmc_hbm_traffic.json), null when not measured for this kernel / batch")
    return roof


def compact_line(line):
    """-> the ONE stdout line: json.dumps of `line` with floats shortened; asserted below LINE_LIMIT."""
    text = json.dumps(_round(line), separators=(",", ":"))
    if len(text) >= LINE_LIMIT:   # cannot happen with the fixed key set; shed the optional parts rather than lose the record
        slim = dict(line)
        for k in ("step_accounting",):
            slim.pop(k, None)
        if isinstance(slim.get("cpu_baseline"), dict):
            slim["cpu_baseline"] = {k: v for k, v in slim["cpu_baseline"].items() if k != "sample"} | \
                {"sample": str(slim["cpu_baseline"].get("sample", ""))[:200]}
        text = json.dumps(_round(slim, 4), separators=(",", ":"))
    assert len(text) < LINE_LIMIT, "bench line is %d bytes" % len(text)
    return text


def write_kernel_table(path, kernels, accounting, header):
    """The per-kernel roofline table (one row per hand-written kernel and per library op / shape) goes to a side file, not
    into the stdout line."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"bench": header, "step_accounting": accounting, "kernels": kernels}, f, indent=1)
    rows = kernels[:int(os.environ.get("MGAR_BENCH_STDERR_ROWS", "16"))]
    log("top kernels of the instrumented step (full table: %s)" % path)
    for r in rows:
        log("  %-44s %4d x %8.3f ms = %8.2f ms  %s" % (r["kernel"][:44], r["launches_per_step"], r["avg_launch_ms"], r["ms_per_step"],
                                                       ("%.0f %s = %.2f of peak" % (r["achieved"], r["unit"], r["frac"])) if "frac" in r else ""))


def spawn_command(n_gpus, argv, port=None):
    """`python bench.py --gpus N` outside a launcher: the command that starts the N ranks (one per GPU) as a CHILD process --
    never an exec, and before torch / the package / anything that touches the GPU is imported in this process."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


class StubStep:
    """MGAR_BENCH_STUB_STEP=1: protocol rehearsal WITHOUT the model (tests/test_bench_cpu.py, gloo on CPU).  It has the
    collective behaviour of TrainStep -- one gradient all-reduce per run() / run_eager() -- and counts what it issues; its
    line is tagged metric "protocol_rehearsal" / data "stub" and is never a measurement."""

    class _M:
        overlap_branches = False

    def __init__(self, device, ddp):
        self.module, self.graph, self.ddp = self._M(), None, ddp
        self.flat = torch.ones(1 << 16, device=device)
        self.collectives = 0

    def _exchange(self):
        if self.ddp:
            import torch.distributed as dist
            dist.all_reduce(self.flat)
            self.flat.div_(dist.get_world_size())
            self.collectives += 1

    def run(self, batch):
        self._exchange()

    run_eager = run


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves (child process; its rank 0 prints the line to our stdout)
        import subprocess
        cmd = spawn_command(args.gpus, sys.argv[1:])
        log("starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    _import_torch()
    stub = bool(os.environ.get("MGAR_BENCH_STUB_STEP"))
    # MGAR_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal of the N > 1 path on a one-GPU box; RCCL
    # refuses two ranks on one device); the real runs use nccl (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("MGAR_BENCH_BACKEND", "nccl")
    if not torch.cuda.is_available() and not (stub and backend == "gloo"):
        raise SystemExit("bench.py needs a GPU: the HIP kernels have no CPU fallback")
    if torch.cuda.is_available():
        if backend != "nccl":
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    ddp = world > 1
    dist = None
    if ddp:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert args.clips % world == 0, "global clip batch must divide over the ranks"
    clips_local = args.clips // world

    def barrier():
        if ddp:
            dist.barrier()
        _sync()

    if stub:
        step, batch, W = StubStep(dev, ddp), None, None
        use_graph = False
    else:
        from multimodal_gar_amd import workload as W
        torch.backends.cudnn.benchmark = not args.no_miopen_find   # MIOpen find mode for the I3D convolutions
        log("building model (rank %d/%d, %d clips on this rank)" % (rank, world, clips_local))
        use_graph = not args.no_graph and not args.ddp_wrapper
        if use_graph and args.route == "voxel" and not W.voxel_route_capturable():
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
        step.module.geometry_prefetch = bool(args.prefetch_geometry) and args.mode == "train"
        step.module.rgb_prefetch = bool(args.prefetch_rgb) and args.mode == "train" and step.module.overlap_branches
        batch = W.make_batch(100 + rank, clips_local, args.frames, args.actors, args.points, args.height, args.width, dev)
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
        step.run(batch); _sync(); log("warmup step %d done" % i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step.run(batch)
        if rank == 0:
            log("timed step %d issued" % i)   # host-side only: no sync inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if not stub:
        log("timed region: %.1f ms/step, peak mem %.1f GB" % (elapsed / args.steps * 1e3, torch.cuda.max_memory_allocated() / 2 ** 30))
    if ddp:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = args.clips * args.steps / elapsed

    roof, kernels, cpu, accounting, kernels_file = None, None, None, None, None
    if not args.no_kernel_timing:
        # every rank issues the same two extra eager steps (their gradient exchanges are collective); rank 0 instruments its second
        rec = extra_steps(step, batch, instrument=(rank == 0 and not stub))
        if rec is not None:
            kernels, accounting = kernel_rooflines(*rec, clips_local, args.precision)
            roof = roofline_entry(kernels)
            log("dominant kernel of the step: %s, %.2f ms/step in %d launches, %.0f %s (%.1f %% of peak)"
                % (roof["kernel"], roof["ms_per_step"], roof["launches_per_step"], roof["achieved"], roof["unit"], 100 * roof["frac"]))
            log("step accounting (ms): %s" % {k: (round(v, 1) if isinstance(v, float) else v) for k, v in accounting.items()})
        barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "train" and not stub:
        cpu = cpu_baseline(args)
    if stub:
        counts = [step.collectives]
        if ddp:
            c = torch.tensor([step.collectives], device=dev, dtype=torch.int64)
            got = [torch.zeros_like(c) for _ in range(world)]
            dist.all_gather(got, c)
            counts = [int(g.item()) for g in got]
        if rank == 0:
            print(compact_line({"metric": "protocol_rehearsal", "value": value, "unit": "stub steps/sec", "n_gpus": world,
                                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "data": "stub",
                                "collectives_per_rank": counts}), flush=True)
        if ddp:
            dist.destroy_process_group()
        return
    if rank == 0:
        what = "fwd+bwd" if args.mode == "train" else "fwd"
        workload = ("%s: %d clips x %d frames x %d actors x %d pts, %dx%d RGB, %s %s, LiDAR route %s, GAT %s"
                    % (args.config, args.clips, args.frames, args.actors, args.points, args.height, args.width, args.precision,
                       "fwd+bwd+Adam" if args.mode == "train" else
                       "train-mode forward (feature payloads + GEMMs bf16; xyz, distances, indices, BN statistics fp32/int32)"
                       if args.precision == "bf16" else "train-mode forward", args.route, "off" if args.no_gat else "on"))
        line = {
            "metric": "clips/sec (%s) at %d actors x %dk pts x %d frames" % (what, args.actors, args.points // 1024, args.frames),
            "value": value, "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": workload, "global_clips": args.clips, "clips_per_gpu": clips_local, "parallelism": "dp%d" % world,
                       "launch": "hip_graph" if step.graph is not None else "eager",
                       "geometry": "prefetched one step ahead (input pipelining)" if args.prefetch_geometry else "in step",
                       "frozen_rgb": "prefetched one step ahead (under the backward)" if args.prefetch_rgb else "in step",
                       "gradient_exchange": "none" if world == 1 else ("ddp_bucketed" if args.ddp_wrapper else "flat_allreduce"),
                       "trainable_params": W.trainable_parameter_count(step.module)},
            "roofline": roof, "cpu_baseline": cpu, "step_accounting": accounting,
        }
        if kernels is not None:
            kernels_file = args.kernels_out or os.path.join(
                ROOT, "gpurun_out", "bench_kernels_%s_%s_%dgpu_%dclips.json" % (args.config, args.route, world, clips_local))
            header = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config")}
            try:
                write_kernel_table(kernels_file, kernels, accounting, header)
                line["kernels_file"] = os.path.relpath(kernels_file, ROOT)
            except OSError as e:   # a read-only tree must not cost the record
                log("could not write the kernel table (%s)" % e)
        print(compact_line(line), flush=True)   # the LAST stdout line, < 4 KB
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
