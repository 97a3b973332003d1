"""Summarise the rocprofv3 passes of `tools/bench_input_prep.py --profile` for image_resize_normalize_kernel:
    python tools/ip_pmc_summary.py <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv>
Counters are KiB (cdna_hip_programming.md); reads here are 4-byte words per lane, so FETCH_SIZE is reported uncorrected."""
import csv
import json
import sys


def counter(path, name):
    vals = [float(r["Counter_Value"]) * 1024.0 for r in csv.DictReader(open(path))
            if r["Counter_Name"] == name and "image_resize_normalize_kernel" in r["Kernel_Name"]]
    return sum(vals) / max(len(vals), 1), len(vals)


def main():
    row = [r for r in csv.DictReader(open(sys.argv[1])) if "image_resize_normalize_kernel" in r["Name"]][0]
    fetch, n = counter(sys.argv[2], "FETCH_SIZE")
    write, _ = counter(sys.argv[3], "WRITE_SIZE")
    alg = 15 * (480 * 3760 * 3 + 720 * 1280 * 3 * 4)
    print(json.dumps({"kernel": "image_resize_normalize_kernel<0, true>", "launches": int(row["Calls"]),
                      "avg_launch_us": float(row["AverageNs"]) / 1e3, "algorithmic_bytes": alg,
                      "achieved_GBps": alg / float(row["AverageNs"]), "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                      "traffic_over_algorithmic": (fetch + write) / alg, "pmc_launches": n}))


if __name__ == "__main__":
    main()
