"""Python counterpart of the reference's ``deltaRice.h5`` module (src/h5.pyx:27,55-61):
exports ``H5FILTER`` and ``register_h5_filter()``.  The filter itself is the C plugin
(deltarice_amd/plugin/libh5deltarice.so) whose callback runs the HIP codec.

Unlike the reference module, importing this one has no side effect: registration is
explicit, because it needs to know which libhdf5 the process uses (h5py's own, found
through ``h5py.h5z.__file__`` exactly as src/h5.pyx:36-53 does, or one you name).
"""
from __future__ import annotations

import ctypes as C
import os

from ._lib import PLUGIN_PATH

H5FILTER = 32025

_plugin = None


def plugin() -> C.CDLL:
    global _plugin
    if _plugin is None:
        if not os.path.exists(PLUGIN_PATH):
            raise ImportError(f"{PLUGIN_PATH} is missing: build it with `make plugin`")
        _plugin = C.CDLL(PLUGIN_PATH, mode=C.RTLD_GLOBAL)
        _plugin.init_filter.argtypes = [C.c_char_p]
        _plugin.init_filter.restype = C.c_int
        _plugin.deltarice_register_h5filter.restype = C.c_int
    return _plugin


def register_h5_filter(libhdf5: str | None = None) -> None:
    """H5Zregister(H5Z_DELTARICE) in the given libhdf5 (default: h5py's, else the process image)."""
    p = plugin()
    libs = [libhdf5] if libhdf5 else []
    if not libs:
        try:
            from h5py import h5d, h5fd, h5s, h5t, h5p, h5z, defs  # same probing order as src/h5.pyx:36-42
            libs = [m.__file__ for m in (h5d, h5fd, h5s, h5t, h5p, h5z, defs)]
        except ImportError:
            libs = []
    for lib in libs:
        if p.init_filter(lib.encode()) == 0:
            break
    ret = p.deltarice_register_h5filter()
    if ret < 0:
        raise RuntimeError("Failed to register DeltaRice HDF5 filter.", ret)
