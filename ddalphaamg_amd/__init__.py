"""ddalphaamg_amd -- MI355X (gfx950) native implementation of the DDalphaAMG V-cycle hot path.

The product is the C-ABI shared library ``libddamg_hip.so`` (hand-written HIP kernels + C++ host
orchestration, sources in ``csrc/``, interface in ``include/ddamg_hip.h`` and the reference's own
``include/dd_alpha_amg.h``).  This Python package is only a thin ctypes mirror of that interface
for tests and benchmarks.  There is no CPU fallback: loading fails loudly if the library is
missing, and every compute entry point fails if no HIP device is visible.
"""
from .api import (Params, Context, Vector, DDAMGError, load_library, library_path,
                  declared_symbols)

__all__ = ["Params", "Context", "Vector", "DDAMGError", "load_library", "library_path", "declared_symbols"]
