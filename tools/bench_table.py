"""Markdown per-kernel table of a bench.py JSON line (the one in profiles/README.md):  python tools/bench_table.py <bench.json>"""
import json
import sys


def main():
    d = json.load(open(sys.argv[1]))
    print("| kernel | launches | ms/step | achieved | of peak | PMC traffic / algorithmic |")
    print("|---|---|---|---|---|---|")
    rows = d["kernels"]
    hw = sorted([r for r in rows if r.get("class") == "hand_written"], key=lambda r: -r["ms_per_step"])
    lib = sorted([r for r in rows if r.get("class") in ("conv", "gemm") and "remaining" not in r["kernel"]], key=lambda r: -r["ms_per_step"])[:6]
    for r in hw + lib:
        alg, ms = r.get("algorithmic_bytes_per_launch"), r["avg_launch_ms"]
        gbs = alg / ms / 1e6 if alg else None
        if "pair_evals_per_s" in r:
            ach, peak = "%.1e pair tests/s" % r["pair_evals_per_s"], "VALU %.0f %%" % (100 * r["valu_frac"])
        elif r.get("bound") == "mfma":
            ach = "%.0f TFLOP/s" % r["achieved"] + (", %.0f GB/s" % gbs if gbs else "")
            peak = "%.0f %% of %.1f TF" % (100 * r["frac"], r["peak"]) + (", %.0f %% HBM" % (gbs / 80) if gbs else " (fp32 MFMA)")
        else:
            ach, peak = "%.0f GB/s" % r["achieved"], "%.0f %%" % (100 * r["frac"])
        tr = "%.2f" % (r["traffic"] / alg) if r.get("traffic") and alg and "pair_evals_per_s" not in r else "—"
        print("| `%s` | %d | %.2f | %s | %s | %s |" % (r["kernel"], r["launches_per_step"], r["ms_per_step"], ach, peak, tr))
    a = d["step_accounting"]
    print("\nstep_accounting: hand-written %.1f ms, library convolutions %.1f ms, library GEMMs %.1f ms, elementwise / copies / reductions "
          "%.1f ms = %.1f ms" % (a["hand_written_kernels_ms"], a["library_conv_ms"], a["library_gemm_ms"], a["torch_elementwise_copy_reduce_ms"], a["sum_ms"]))


if __name__ == "__main__":
    main()
