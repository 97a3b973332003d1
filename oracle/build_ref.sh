#!/bin/bash
# Builds oracle/_ref/roiaware_pool3d_ref.so FROM THE REFERENCE'S OWN SOURCE FILE, where it lies:
#   /root/reference/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp
# -- a torch C++ extension whose points_in_boxes_cpu() (lines 118-168) is the only CPU implementation of a hot-path-adjacent
# op the reference ships.  g++ on that one file against the torch / Python headers of this image; nothing is copied, no
# stand-in is written.  The file also declares the CUDA launchers of its .cu sibling (not buildable here: no nvcc); they
# stay undefined symbols of the shared object and are never called -- the loader binds lazily (tests open the module with
# RTLD_LAZY) and only points_in_boxes_cpu is used, to pin oracle/mgar_oracle.c::orc_points_in_boxes' box test.
# Test infrastructure only; runs in the build container only (the GPU box has no /root/reference and uses the built file).
set -e
SRC=/root/reference/pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp
OUT="$(dirname "$0")/_ref"
[ -f "$SRC" ] || { echo "reference tree not present: skipping oracle/_ref"; exit 0; }
mkdir -p "$OUT"
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
TORCH=$(python3 -c "import torch, os; print(os.path.dirname(torch.__file__))")
g++ -O2 -fPIC -shared -std=c++17 -w -DTORCH_EXTENSION_NAME=roiaware_pool3d_ref -DTORCH_API_INCLUDE_EXTENSION_H \
    -I"$PYINC" -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" \
    "$SRC" -o "$OUT/roiaware_pool3d_ref.so" \
    -L"$TORCH/lib" -ltorch -ltorch_cpu -lc10 -ltorch_python -Wl,-rpath,"$TORCH/lib" -Wl,--unresolved-symbols=ignore-all
echo "built $OUT/roiaware_pool3d_ref.so"
