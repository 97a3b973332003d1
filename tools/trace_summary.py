"""Summarise the LAST optimizer step of a rocprofv3 --kernel-trace CSV of bench.py.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d out -o run --output-format csv -- python bench.py --steps 1 --warmup 1 ...
    python tools/trace_summary.py out/run_kernel_trace.csv [--csv profiles/rNN_timed_step_kernels.csv] [--long 0.7]

Prints span / busy / per-stream totals, a per-category split and the top kernels of the timed step
(the kernels between the last two optimizer launches), optionally writing the per-kernel table."""
import argparse
import collections
import csv


def category(n):
    if n.startswith("Cijk"):
        return "library GEMM"
    if "_ZN2ck" in n or "Im3d2Col" in n or "batched_transpose" in n:
        return "library conv (MIOpen/CK)"
    if "mgar::conv3d_wino" in n or "mgar::stem_conv3d" in n:
        return "mgar convolution (fp32 MFMA)"
    if "mgar::bn_" in n:
        return "mgar bn_act"
    if "mgar::pointwise_" in n or "mgar::rowmajor_dw" in n:
        return "mgar pointwise MFMA"
    if "mgar::" in n:
        return "mgar irregular ops"
    if "at::" in n or "rocprim" in n or "rocclr" in n:
        return "torch elementwise / copies"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--csv")
    ap.add_argument("--long", type=float, default=0.0, help="also list, in time order, kernels longer than this many ms")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "multi_tensor" in r["Kernel_Name"].lower()]
    groups = []
    for i in opt:
        if not groups or i - groups[-1][-1] > 50:
            groups.append([i])
        else:
            groups[-1].append(i)
    step = rows[groups[-2][-1] + 1:groups[-1][-1] + 1] if len(groups) >= 2 else rows
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6  # noqa: E731
    t0 = int(step[0]["Start_Timestamp"])
    span = (max(int(r["End_Timestamp"]) for r in step) - t0) / 1e6
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    total = sum(dur(r) for r in step)
    print("timed step: %d kernels, span %.1f ms, GPU busy %.1f ms, sum of kernel time %.1f ms" % (len(step), span, busy / 1e6, total))
    by_stream = collections.defaultdict(float)
    cat = collections.defaultdict(float)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in step:
        by_stream[r["Stream_Id"]] += dur(r)
        cat[category(r["Kernel_Name"])] += dur(r)
        k = agg[r["Kernel_Name"]]
        k[0] += dur(r)
        k[1] += 1
    print("per stream:", {k: round(v, 1) for k, v in by_stream.items()})
    for k, v in sorted(cat.items(), key=lambda x: -x[1]):
        print("  %8.2f ms  %5.1f %%  %s" % (v, 100 * v / total, k))
    print("top kernels:")
    for n, v in sorted(agg.items(), key=lambda x: -x[1][0])[:a.top]:
        print("  %8.2f ms %5d x  %s" % (v[0], v[1], n[:110]))
    if a.long:
        print("kernels longer than %.2f ms, in time order:" % a.long)
        for r in step:
            if dur(r) > a.long:
                print("  %7.1f  %6.2f  s%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, dur(r), r["Stream_Id"], r["Kernel_Name"][:100]))
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
            for n, v in sorted(agg.items(), key=lambda x: -x[1][0]):
                w.writerow([n, v[1], int(v[0] * 1e6), int(v[0] * 1e6 / v[1]), "%.2f" % (100 * v[0] / total)])


if __name__ == "__main__":
    main()
