"""indexr_amd -- MI355X (gfx950) decoder for IndexR VLT column packs.

The product is the C-ABI shared library `libixrgpu.so` (include/ixr_gpu.h).  This package only loads it (ctypes) and adds
thin helpers for tests and bench.py.  Importing `indexr_amd.lib` raises if the library has not been built.
"""
__all__ = ["lib"]
