"""HBM traffic per launch of the hand-written kernels from two rocprofv3 PMC passes of bench.py.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_fetch -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_write -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing
    python tools/pmc_traffic.py out_fetch/p_counter_collection.csv out_write/p_counter_collection.csv [clips_per_gpu=8] > profiles/pmc_hbm_traffic.json

Counters are in KiB (/opt/skills/guides/cdna_hip_programming.md: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024) and are
collected in SEPARATE passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  gfx950 correction
(MI355X_MICROARCH.md, section HBM): FETCH_SIZE reports exactly half of the bytes of a wide (16 B / lane)
coalesced streaming read, so the read side of the kernels listed in WIDE_READERS -- whose global reads
are all float4 -- is doubled; WRITE_SIZE is exact for 16-byte stores and float atomics.  Kernels that
read through 4-byte gathers are reported uncorrected and flagged "uncalibrated".  Both steps of the run
(warm-up + timed) execute the same launches, so the average is over all launches of a kernel."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# ktimer name (csrc/errors.hip) -> substring of the kernel symbol
KERNELS = {
    "bn_partial_kernel": "mgar::bn_partial_kernel", "bn_apply_kernel": "mgar::bn_apply_kernel", "bn_max_vec_kernel": "mgar::bn_max_vec_kernel",
    "bn_bwd_partial_kernel": "mgar::bn_bwd_partial_kernel", "bn_bwd_apply_kernel": "mgar::bn_bwd_apply_kernel",
    "bn_max_bwd_partial_kernel": "mgar::bn_max_bwd_partial_kernel", "bn_max_bwd_apply_kernel": "mgar::bn_max_bwd_apply_kernel",
    "pointwise_fwd_kernel": "mgar::pointwise_fwd_kernel", "pointwise_dw_kernel": "mgar::pointwise_dw_kernel<",
    "rowmajor_dw_kernel": "mgar::rowmajor_dw_kernel", "maxpool3d_same_kernel": "mgar::maxpool3d_same",
    "fps_kernel": "mgar::fps_", "ball_query_kernel": "mgar::ball_query_kernel", "ball_query_grid_kernel": "mgar::ball_query_grid_kernel",
    "three_nn_grid_kernel": "mgar::three_nn_grid_kernel", "point_grid_build": "mgar::pg_", "spconv_gemm": "mgar::spconv_os_kernel",
    "spconv_dw": "mgar::spconv_pairs_dw_kernel", "spconv_index": "mgar::sp_", "gatv2_bwd": "mgar::gatv2_bwd_", "gatv2_fwd": "mgar::gatv2_fwd_kernel",
    "dafm_attn_fwd": "mgar::dafm_fwd_kernel", "dafm_attn_bwd": "mgar::dafm_bwd_", "roi_align_fwd": "mgar::roi_align_fwd_kernel",
    "roi_align_bwd": "mgar::roi_align_bwd_kernel", "voxel_query_kernel": "mgar::voxel_query_kernel", "three_nn_kernel": "mgar::three_nn_kernel",
    "three_interp_fwd": "mgar::three_interp_batch_fwd", "three_interp_bwd": "mgar::three_interp_batch_bwd",
    "query_group_fwd": "mgar::qg_", "query_group_bwd": "mgar::qg_", "query_group_inverse_index": "mgar::qg_inv_",
    "stem_conv3d_kernel": "mgar::stem_conv3d_", "conv3d_wino_kernel": "mgar::conv3d_wino_kernel", "voxel_roi_pool_fwd": "mgar::vrp_fwd_kernel", "voxel_roi_pool_bwd": "mgar::vrp_bwd_kernel",
}
WIDE_READERS = {"bn_partial_kernel", "bn_apply_kernel", "bn_max_vec_kernel", "bn_bwd_partial_kernel", "bn_bwd_apply_kernel",
                "bn_max_bwd_apply_kernel", "pointwise_fwd_kernel", "pointwise_dw_kernel", "maxpool3d_same_kernel"}
# (stem_conv3d_kernel reads its input patches with 4-byte loads: uncalibrated, like the gather kernels)


def per_kernel(path, counter):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        for key, sub in KERNELS.items():
            if sub in n and not (key == "query_group_bwd" and "bwd" not in n) and not (key == "query_group_fwd" and "_fwd_kernel" not in n):
                t = tot[key]
                t[0] += float(r["Counter_Value"]) * 1024.0
                t[1] += 1
                break
    return tot


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"unit": "bytes per launch (HBM-side, FETCH_SIZE [x2 for float4 streaming readers] + WRITE_SIZE, KiB * 1024)",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1 at config c3",
           "clips_per_gpu": int(sys.argv[3]) if len(sys.argv) > 3 else 8,   # bench.py attaches the figures only to this per-rank batch
           "commit": sys.argv[4] if len(sys.argv) > 4 else "unknown",
           # sha256[:16] of the .hip file each kernel was compiled from at measurement time: bench.py attaches a figure only
           # while the source is unchanged
           "per_launch_bytes": {}, "source_sha16": {}, "detail": {}}
    from multimodal_gar_amd.op_timer import source_sha16
    for k in KERNELS:
        if k not in fetch or k not in write:
            continue
        f = fetch[k][0] / fetch[k][1]
        w = write[k][0] / write[k][1]
        wide = k in WIDE_READERS
        total = (2.0 * f if wide else f) + w
        out["per_launch_bytes"][k] = round(total)
        out["source_sha16"][k] = source_sha16(k)
        out["detail"][k] = {"launches": fetch[k][1], "fetch_size_bytes_raw": round(f), "write_size_bytes": round(w),
                            "read_correction": "x2 (16 B/lane streaming reads)" if wide else "none (uncalibrated: 4-byte gathers / scalar streams)"}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
