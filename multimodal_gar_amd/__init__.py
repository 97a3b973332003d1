"""MGAR-net hot path on MI355X (gfx950): hand-written HIP kernels behind a C ABI (csrc/, include/mgar_ops.h)
and the host-side mirror of the reference's model / pcdet interfaces."""
import os as _os

# MIOpen's user find-db for the I3D convolution shapes on gfx950 (consulted when the caller enables
# torch.backends.cudnn.benchmark); must be in the environment before MIOpen initialises.
_os.environ.setdefault("MIOPEN_USER_DB_PATH", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db"))
