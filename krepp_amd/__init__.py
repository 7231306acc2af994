"""krepp_amd — MI355X-native implementation of krepp's per-read `dist` query path.

The package holds only what that path needs: ``csrc/`` (HIP kernels + the C ABI declared
in ``include/krepp_amd.h`` + the C++ host side: index reader, FASTX batcher, CLI) and
``capi`` (a ctypes mirror of the C ABI used by tests and bench.py).  Importing the
package does not load the native library; ``capi.load()`` does, and raises if the HIP
extension has not been built.
"""
from . import capi  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["capi", "synth"]
