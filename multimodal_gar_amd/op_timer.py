"""Per-op device timing of everything that is NOT a hand-written kernel (diagnostic for bench.py's roofline table).

The hand-written kernels are timed by the library itself (mgar_ktimer_*, csrc/errors.hip).  The rest of a step is
PyTorch-ROCm library work -- MIOpen / CK convolutions of the I3D trunk, hipBLASLt / Tensile GEMMs, elementwise and copy
kernels.  ``AtenOpTimer`` is a TorchDispatchMode that brackets every aten op that launches device work with two events
on the current stream and, for convolutions and GEMMs, computes the op's FLOPs with torch.utils.flop_counter's formulas
from the actual argument shapes, so that every library op gets a measured TFLOP/s against the MFMA peak of its dtype
and the WHOLE step is accounted for (52 % of the step had no roofline row).

Only meaningful for an eager, single-stream step (as bench.py issues its instrumented step): the time between an op's
two events is then the time of the kernels that op launched.
"""
import os

import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils.flop_counter import flop_registry

MFMA_PEAK_TFLOPS = {torch.float32: 157.3, torch.bfloat16: 2500.0, torch.float16: 2500.0}   # MI355X_MICROARCH.md, dense
_CONV = ("convolution", "_convolution", "miopen_convolution", "cudnn_convolution", "convolution_backward")
_GEMM = ("mm", "addmm", "bmm", "baddbmm", "_scaled_mm")
# allocate-only / metadata ops: they launch nothing, two events around them would only measure the event overhead
_NO_KERNEL = frozenset(("empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "resize_", "set_", "detach",
                        "alias", "_unsafe_view", "lift_fresh", "is_same_size", "_local_scalar_dense", "item", "sym_size",
                        "sym_numel", "sym_stride", "record_stream", "is_pinned", "_pin_memory"))


_ALL_SHAPES = bool(os.environ.get("MGAR_OPTIMER_ALL_SHAPES"))   # diagnostics: one row per (op, shapes) for every aten op


def _shapes(args):
    out = []
    for a in args:
        if torch.is_tensor(a):
            out.append("x".join(str(int(s)) for s in a.shape))
    return ",".join(out[:3])


class AtenOpTimer(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.records = []      # (op name, class, key, flops, dtype, ev0, ev1)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if getattr(func, "is_view", False) or func.__name__.split(".")[0] in _NO_KERNEL:
            return func(*args, **kwargs)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        out = func(*args, **kwargs)
        ev1.record()
        packet = getattr(func, "overloadpacket", None)
        name = func.__name__.split(".")[0] if hasattr(func, "__name__") else str(func)
        flops, cls, key, dtype = 0, "other", "", None
        if packet in flop_registry and (name in _CONV or name in _GEMM):
            try:
                flops = int(flop_registry[packet](*args, **kwargs, out_val=out))
            except Exception:   # noqa: BLE001 -- a formula that cannot digest the arguments: keep the time, drop the flops
                flops = 0
            cls = "conv" if name in _CONV else "gemm"
            key = _shapes(args)
            for a in args:
                if torch.is_tensor(a) and a.is_floating_point():
                    dtype = a.dtype
                    break
        if not key and _ALL_SHAPES:
            key = _shapes(args)
        self.records.append((name, cls, key, flops, dtype, ev0, ev1))
        return out

    def table(self, top=12):
        """-> (rows, totals): rows = per (op, shapes) for conv / gemm ops (largest total time first, `top` of each class)
        plus one row per remaining aten op name; totals = {class: ms}."""
        torch.cuda.synchronize()
        agg, totals = {}, {"conv": 0.0, "gemm": 0.0, "other": 0.0}
        for name, cls, key, flops, dtype, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            totals[cls] += ms
            k = (cls, name, key, dtype)
            r = agg.setdefault(k, [0, 0.0, 0])
            r[0] += 1; r[1] += ms; r[2] += flops
        rows = []
        for (cls, name, key, dtype), (calls, ms, flops) in agg.items():
            row = {"kernel": "lib:aten.%s%s" % (name, (" [" + key + "]") if key else ""), "class": cls, "launches_per_step": calls,
                   "ms_per_step": ms, "avg_launch_ms": ms / calls}
            if cls != "other" and flops and ms > 0:
                peak = MFMA_PEAK_TFLOPS.get(dtype, 157.3)
                tf = flops / ms / 1e9
                row.update({"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                            "flops_per_launch": flops / calls, "dtype": str(dtype).replace("torch.", ""), "traffic": None})
            rows.append(row)
        rows.sort(key=lambda r: -r["ms_per_step"])
        keep, seen = [], {"conv": 0, "gemm": 0, "other": 0}
        rest = {"conv": [0, 0.0, 0.0], "gemm": [0, 0.0, 0.0], "other": [0, 0.0, 0.0]}
        for r in rows:
            c = r["class"]
            if seen[c] < top:
                keep.append(r); seen[c] += 1
            else:
                rest[c][0] += r["launches_per_step"]; rest[c][1] += r["ms_per_step"]
                rest[c][2] += r.get("flops_per_launch", 0.0) * r["launches_per_step"]
        for c, (calls, ms, flops) in rest.items():
            if calls:
                row = {"kernel": "lib:%s (all remaining shapes)" % c, "class": c, "launches_per_step": calls, "ms_per_step": ms,
                       "avg_launch_ms": ms / calls}
                if flops and ms > 0:
                    row.update({"bound": "mfma", "achieved": flops / ms / 1e9, "unit": "TFLOP/s"})
                keep.append(row)
        return keep, totals


# ---- which source file a timed kernel comes from (bench.py drops PMC traffic figures measured on an older version) ----
KERNEL_SOURCES = {
    "bn_partial_kernel": "bn_act.hip", "bn_apply_kernel": "bn_act.hip", "bn_max_vec_kernel": "bn_act.hip",
    "bn_bwd_partial_kernel": "bn_act.hip", "bn_bwd_apply_kernel": "bn_act.hip", "bn_max_bwd_partial_kernel": "bn_act.hip",
    "bn_max_bwd_apply_kernel": "bn_act.hip", "pointwise_fwd_kernel": "pointwise_fwd.hip", "pointwise_dw_kernel": "pointwise_dw.hip",
    "rowmajor_dw_kernel": "rowmajor_dw.hip", "maxpool3d_same_kernel": "maxpool3d.hip", "fps_kernel": "fps.hip",
    "ball_query_kernel": "ball_query.hip", "three_nn_kernel": "interpolate.hip", "three_interp_fwd": "interpolate.hip",
    "three_interp_bwd": "interpolate.hip", "query_group_fwd": "query_group.hip", "query_group_bwd": "query_group.hip",
    "voxel_roi_pool_fwd": "voxel_roi_pool.hip", "voxel_roi_pool_bwd": "voxel_roi_pool.hip", "stem_conv3d_kernel": "stem_conv.hip",
    "query_group_inverse_index": "query_group.hip", "image_resize_normalize": "input_prep.hip",
    "ball_query_grid_kernel": "ball_query_grid.hip", "three_nn_grid_kernel": "ball_query_grid.hip", "point_grid_build": "ball_query_grid.hip",
    "spconv_gemm": "sparse_conv.hip", "spconv_dw": "sparse_conv.hip", "spconv_index": "sparse_conv.hip", "gatv2_fwd": "gatv2.hip",
    "gatv2_bwd": "gatv2.hip", "dafm_attn_fwd": "dafm.hip", "dafm_attn_bwd": "dafm.hip", "roi_align_fwd": "roi_align.hip",
    "roi_align_bwd": "roi_align.hip", "voxel_query_kernel": "voxel_query.hip", "conv3d_wino_kernel": "conv3d_wino.hip",
}


def source_sha16(kernel):
    """sha256[:16] of the .hip file `kernel` is compiled from (None if unknown)."""
    import hashlib
    import os
    f = KERNEL_SOURCES.get(kernel)
    if f is None:
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", f)
    if not os.path.exists(path):
        return None
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
