"""deltarice_amd -- MI355X-native Delta-Rice codec behind HDF5 filter 32025.

Layout: ``csrc/`` holds the HIP kernels, the C ABI and the H5Z plugin source;
``codec`` is the host-side batch API over device-resident chunks; ``h5`` mirrors
the reference's ``deltaRice.h5`` registration module; ``dist`` shards a batch of
chunks over the GPUs of a node; ``optimise`` searches RiceParameter and encoding filter
(docs/Optimization.md of the reference).  There is no CPU implementation in this package.
"""
from ._lib import DeltaRiceError, LIB_PATH, PLUGIN_PATH  # noqa: F401

H5FILTER = 32025


def __getattr__(name):
    # torch is only needed for the device-resident API; keep `import deltarice_amd` light
    if name in ("Context", "Plan", "EncodedBatch", "parse_opts"):
        from . import codec
        return getattr(codec, name)
    raise AttributeError(name)
