"""speedy.f90_amd -- MI355X-native grid<->spectral transform path for speedy.f90.

Only what the hot path needs lives here:
    csrc/      HIP kernels (gfx950) + the C-ABI (include/spdy.h) + host table generation
    fortran/   ISO_C_BINDING drop-in for the reference's `spectral` module (the real host)
    spectral.py  Python mirror of that module over the same C-ABI (ctypes), used by tests,
                 smoke() and bench.py

The directory name contains a dot, so import it through the repo-root shim
``import speedy_f90_amd`` (speedy_f90_amd.py).
"""
from ._lib import LIB_PATH, SpdyError, build, load  # noqa: F401
from .spectral import RESOLUTIONS, Graph, Spectral, check  # noqa: F401
from . import sharding  # noqa: F401
