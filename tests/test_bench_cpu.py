"""bench.py's contract, checked without a GPU: the stdout line stays small enough for the driver's stdout window, `--gpus N`
starts N ranks by itself, and every rank issues the same number of collectives."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports neither torch nor the package at module level)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _fake_rows(n):
    rows = []
    for i in range(n):
        rows.append({"kernel": "some_rather_long_kernel_name_%02d<float, 16, true>" % i, "class": "hand_written" if i % 2 else "conv",
                     "launches_per_step": 120 + i, "ms_per_step": 23.123456789 - 0.1 * i, "avg_launch_ms": 0.19269547,
                     "algorithmic_bytes_per_launch": 1.234567e9, "flops_per_launch": 2.446e12, "traffic": 9.23e9 if i == 1 else None,
                     "traffic_source": "profiles/pmc_hbm_traffic.json@abcdef (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                     "bound": "mfma", "achieved": 104.8123456, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.66634567,
                     "mfma_dtype": "f32", "hbm_gbs": 512.123, "hbm_frac": 0.064})
    return rows


def test_line_is_compact_and_complete():
    kernels = _fake_rows(58)                                   # the table that made BENCH_r02's line 22 KB
    roof = bench.roofline_entry(kernels)
    assert roof["kernel"] == kernels[0]["kernel"] and roof["dominant_hand_written"]["kernel"] == kernels[1]["kernel"]
    assert roof["dominant_hand_written"]["traffic"] == 9.23e9 and "measured_in" in roof
    line = {
        "metric": "clips/sec (fwd+bwd) at 32 actors x 16k pts x 15 frames", "value": 35.512345678, "unit": "clips/sec", "n_gpus": 1,
        "steps": 20, "warmup": 5, "ms_per_step": 225.2812345, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "c3: 8 clips x 15 frames x 32 actors x 16384 pts, 720x1280 RGB, fp32 fwd+bwd+Adam, LiDAR route pointnet2, GAT on",
                   "global_clips": 8, "clips_per_gpu": 8, "parallelism": "dp1", "launch": "hip_graph", "gradient_exchange": "none",
                   "trainable_params": 29700000},
        "roofline": roof,
        "cpu_baseline": {"value": 0.0826, "unit": "clips/sec", "cores": 16, "kind": "port", "timed_passes": 3, "sample": "x" * 330},
        "step_accounting": {"hand_written_kernels_ms": 114.0, "library_conv_ms": 63.0, "library_gemm_ms": 52.0,
                            "torch_elementwise_copy_reduce_ms": 17.0, "sum_ms": 246.0, "instrumented_step_wall_ms": 261.0},
        "kernels_file": "gpurun_out/bench_kernels_c3_pointnet2_1gpu_8clips.json",
    }
    text = bench.compact_line(line)
    assert "\n" not in text and len(text) < bench.LINE_LIMIT == 4096
    got = json.loads(text)
    for k in REQUIRED:
        assert k in got, k
    assert "kernels" not in got
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in got["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert got["config"]["workload"].startswith("c3") and "model" not in got["config"]
    assert abs(got["value"] - 35.512345678) < 1e-3


def test_kernel_table_goes_to_a_side_file(tmp_path):
    path = str(tmp_path / "sub" / "kernels.json")
    bench.write_kernel_table(path, _fake_rows(58), {"sum_ms": 1.0}, {"metric": "m"})
    got = json.load(open(path))
    assert len(got["kernels"]) == 58 and got["step_accounting"] == {"sum_ms": 1.0}


def test_spawn_command_starts_one_rank_per_gpu():
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def _run_bench(argv, extra_env, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)
    return p


@pytest.mark.timeout(600)
def test_gpus_2_without_launcher_runs_two_ranks_with_equal_collective_counts():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself (r2: it silently measured one), and the
    control flow around the timed region / the instrumented step must issue the same collectives on every rank (r2: rank 0
    issued one more gradient all-reduce than the others).  Driven end to end through main() with the stub step on gloo."""
    p = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "2"],
                   {"MGAR_BENCH_STUB_STEP": "1", "MGAR_BENCH_BACKEND": "gloo", "OMP_NUM_THREADS": "1"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    got = json.loads(lines[-1])
    assert got["metric"] == "protocol_rehearsal" and got["data"] == "stub" and got["n_gpus"] == 2
    # warmup 2 + timed 3 + the two extra eager steps
    assert got["collectives_per_rank"] == [7, 7]
    assert len(lines[-1]) < 4096


def test_world_size_must_match_gpus():
    p = _run_bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "MGAR_BENCH_STUB_STEP": "1", "MGAR_BENCH_BACKEND": "gloo"})
    assert p.returncode != 0 and "--gpus 1 but the launcher started 2 ranks" in p.stderr


def test_no_gpu_no_fallback():
    p = _run_bench(["--steps", "1"], {})
    assert p.returncode != 0 and "no CPU fallback" in p.stderr
