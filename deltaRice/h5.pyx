# cython: language_level=3
"""``deltaRice.h5`` -- the h5py binding of HDF5 filter 32025, MI355X build.

Same module path, attributes and import-time behaviour as the reference's Cython module
(/root/reference/src/h5.pyx): ``H5FILTER`` (:27), ``register_h5_filter()`` (:55-58), and the filter is
registered with the libhdf5 that h5py itself uses when the module is imported (:32-53, :61), so

    import h5py, deltaRice.h5
    f.create_dataset("x", data=a, compression=deltaRice.h5.H5FILTER, compression_opts=(8, 7000))

keeps working unchanged.  The C side is this repository's plugin library
(deltarice_amd/plugin/libh5deltarice.so, include/deltarice_h5filter.h): the filter callback runs the HIP
codec; there is no CPU codec behind it.  Only ``H5Zregister`` has to be bound late (the reference's
hdf5_dl.c binds eleven functions it never calls, src/hdf5_dl.c:194-267).
"""
import sys

import h5py
from h5py import defs, h5d, h5fd, h5p, h5s, h5t, h5z

cdef extern from "deltarice_h5filter.h":
    enum: H5Z_FILTER_DELTARICE
    int deltarice_register_h5filter()
    int init_filter(const char *libname)

H5FILTER = H5Z_FILTER_DELTARICE


def _h5py_libraries():
    """h5py's own extension modules, in the order the reference probes them (src/h5.pyx:36-42): each of them
    links the libhdf5 instance h5py runs on, so H5Zregister resolved through any of them is the right one."""
    return [m.__file__ for m in (h5d, h5fd, h5s, h5t, h5p, h5z, defs)]


def _bind_hdf5():
    libs = _h5py_libraries()
    for lib in libs:
        if init_filter(lib.encode("utf-8")) == 0:
            return lib
    raise RuntimeError("Failed to load all HDF5 symbols using these libs: {}".format(libs))


def register_h5_filter():
    """H5Zregister(H5Z_DELTARICE); RuntimeError if HDF5 refuses (src/h5.pyx:55-58)."""
    ret = deltarice_register_h5filter()
    if ret < 0:
        raise RuntimeError("Failed to register DeltaRice HDF5 filter.", ret)


if not sys.platform.startswith("win"):
    HDF5_LIBRARY = _bind_hdf5()
register_h5_filter()
