"""``deltaRice.h5`` as users of the reference import it: ``H5FILTER`` (32025) and ``register_h5_filter()``,
and -- like the reference module -- the filter is registered with h5py's libhdf5 when this module is
imported.  The work is done by :mod:`deltarice_amd.h5` (C plugin + HIP codec); importing this module
without h5py installed raises ImportError, as it does for the reference."""
import h5py  # noqa: F401  (the reference module needs h5py's libhdf5 too)

from deltarice_amd.h5 import H5FILTER, register_h5_filter

register_h5_filter()

__all__ = ["H5FILTER", "register_h5_filter"]
