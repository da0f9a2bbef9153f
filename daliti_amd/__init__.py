"""daliti_amd -- MI355X-native scan-to-map registration engine for DaLiTI's eskf_lio.

The product is the C-ABI shared library ``daliti_amd/_lib/libdaliti_s2m.so`` (HIP kernels for
gfx950 + C++ host engine, declared in ``include/daliti_s2m.h``).  This package is a thin ctypes
binding over that ABI for tests, ``bench.py`` and Python callers.  There is no CPU fallback: the
compute entry points raise :class:`S2MError` when the library or a gfx950 device is missing.
"""
from .engine import (Config, Engine, PassOut, IterLog, S2MError, build_library, library_path,  # noqa: F401
                     load_library, ABI_SYMBOLS)

__all__ = ["Config", "Engine", "PassOut", "IterLog", "S2MError", "build_library", "library_path",
           "load_library", "ABI_SYMBOLS"]
