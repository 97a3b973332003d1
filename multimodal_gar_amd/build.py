"""Build recipe for libmgar_hip.so (the gfx950 kernels + C ABI of include/mgar_ops.h).

Run as ``python -m multimodal_gar_amd.build`` or through ``__graft_entry__.build()``.
hipcc cross-compiles for gfx950 without a GPU; the .so is built in-tree (git-ignored)
so that it travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmgar_hip.so")

# -ffp-contract=off: the kernels spell out every fma they want (floating-point
# convention); the compiler must not fuse or split anything else.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
               "-fvisibility=hidden", "-Wall", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "hipcc")
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(HERE, "..", "include", "mgar_ops.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + HIPCC_FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
