import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS_DIR = os.path.dirname(os.path.abspath(__file__))
if TESTS_DIR not in sys.path:
    sys.path.insert(0, TESTS_DIR)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def record_error(what, err, scale, tol):
    """Parity tests call this with every comparison they make: on the GPU box the measured margins land in
    gpurun_out/parity_margins.txt (test id, what, err / scale, asserted tolerance) so that tolerances are set from data."""
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
        with open(os.path.join(out, "parity_margins.txt"), "a") as f:
            f.write("%-110s %-40s rel %.3e  tol %.1e\n" % (test[-110:], str(what)[:40], err / max(scale, 1e-300), tol))
