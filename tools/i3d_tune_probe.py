"""Probe: does an exhaustive MIOpen search (MIOPEN_FIND_ENFORCE) find faster I3D convolutions than the
immediate-mode default?  Prints ms per clip for the frozen I3D forward at c3's 15 x 720 x 1280 input."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_gar_amd.model.backbone import InceptionI3d  # noqa: E402


def main():
    torch.backends.cudnn.benchmark = os.environ.get("PROBE_BENCHMARK", "0") == "1"
    m = InceptionI3d(final_endpoint="Mixed_4f"); m.build(); m = m.cuda().train()
    nb = int(os.environ.get("PROBE_BATCH", "1"))
    x = torch.randn(nb, 3, 15, 720, 1280, device="cuda")
    with torch.no_grad():
        t0 = time.time(); m.extract_features(x); torch.cuda.synchronize()
        print("first pass %.1f s" % (time.time() - t0), flush=True)
        for _ in range(2):
            m.extract_features(x)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            m.extract_features(x)
        torch.cuda.synchronize()
        print("I3D forward, batch %d: %.2f ms / clip" % (nb, (time.time() - t0) / 5 / nb * 1e3), flush=True)


if __name__ == "__main__":
    main()
