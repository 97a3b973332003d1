"""The float64 ground truth of the op-module compositions (oracle/fp64_truth.py) against the fp32 oracle backend on the
CPU: pins the truth's composition (both are restatements of the same reference modules; they must agree to fp32 rounding),
and records how far fp32 itself is from float64 on these train-mode BatchNorm chains -- the yardstick the device test
(tests/test_fp64_truth_gpu.py) holds the HIP path to."""
import pytest

from truth_cases import cases, rel_err

CASES = {c.name: c for c in cases()}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_backend_agrees_with_float64_truth(name):
    c = CASES[name]
    out_o, gi_o, gp_o = c.oracle()
    out_t, gi_t, gp_t = c.truth()
    assert out_o.shape == out_t.shape and len(gi_o) == len(gi_t) and len(gp_o) == len(gp_t) and len(gp_t) >= 2
    assert rel_err(out_o, out_t)[0] < 1e-5, rel_err(out_o, out_t)
    for a, b in zip(gi_o + gp_o, gi_t + gp_t):
        assert a.shape == b.shape
        assert rel_err(a, b)[0] < 1e-4, (name, rel_err(a, b))
