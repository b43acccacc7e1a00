"""Loads the C-ABI library (include/deltarice_hip.h).  No fallback: if the HIP
library is missing or cannot be loaded, importing the codec raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRX_LIB_PATH: load another build of the same ABI (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("DRX_LIB_PATH") or os.path.join(_HERE, "libdeltarice_hip.so")
PLUGIN_PATH = os.path.join(_HERE, "plugin", "libh5deltarice.so")

DRX_OK = 0
STATUS_NAMES = {0: "DRX_OK", 1: "DRX_ERR_ARG", 2: "DRX_ERR_DEVICE", 3: "DRX_ERR_CAPACITY",
                4: "DRX_ERR_CORRUPT", 5: "DRX_ERR_UNSUPPORTED", 6: "DRX_ERR_NOMEM"}
DRX_MAX_TAPS = 64


class DrxOpts(C.Structure):
    _fields_ = [("rice_k", C.c_uint32), ("wave_len", C.c_int64), ("n_taps", C.c_uint32),
                ("taps", C.c_int32 * DRX_MAX_TAPS)]


class DeltaRiceError(RuntimeError):
    def __init__(self, status: int, msg: str = ""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {msg}")


# every symbol include/deltarice_hip.h declares: (restype, argtypes)
_vp, _u64, _u32 = C.c_void_p, C.c_uint64, C.c_uint32
SIGNATURES = {
    "drx_version": (C.c_char_p, []),
    "drx_status_str": (C.c_char_p, [C.c_int]),
    "drx_device_count": (C.c_int, []),
    "drx_parse_cd_values": (C.c_int, [C.c_size_t, C.POINTER(C.c_uint), C.POINTER(DrxOpts)]),
    "drx_ctx_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "drx_ctx_destroy": (None, [_vp]),
    "drx_ctx_synchronize": (C.c_int, [_vp]),
    "drx_ctx_last_error": (C.c_char_p, [_vp]),
    "drx_ctx_stream": (_vp, [_vp]),
    "drx_ctx_device": (C.c_int, [_vp]),
    "drx_ctx_host_staging": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "drx_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "drx_plan_create": (C.c_int, [_vp, _u64, C.POINTER(_u32), C.POINTER(_u32), _u32, C.POINTER(_vp)]),
    "drx_plan_create_uniform": (C.c_int, [_vp, _u64, _u32, _u32, _u32, C.POINTER(_vp)]),
    "drx_plan_set_filter": (C.c_int, [_vp, _u32, C.POINTER(C.c_int32)]),
    "drx_plan_destroy": (None, [_vp]),
    "drx_plan_n_chunks": (_u64, [_vp]),
    "drx_plan_total_samples": (_u64, [_vp]),
    "drx_plan_total_waves": (_u64, [_vp]),
    "drx_plan_max_encoded_words": (_u64, [_vp]),
    "drx_plan_wave_words": (_vp, [_vp]),
    "drx_plan_wave_word_off": (_vp, [_vp]),
    "drx_plan_last_decode_path": (C.c_uint32, [_vp]),
    "drx_plan_last_encode_path": (C.c_uint32, [_vp]),
    "drx_plan_read_wave_words": (C.c_int, [_vp, C.POINTER(_u32)]),
    "drx_encode": (C.c_int, [_vp, _vp, _vp, _u64, _vp]),
    "drx_decode": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "drx_decode_with_wave_words": (C.c_int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "drx_estimate_words": (C.c_int, [_vp, _vp, C.POINTER(_u64)]),
    "drx_plan_last_timings": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "drx_plan_finish": (C.c_int, [_vp, C.POINTER(_u64)]),
    "drx_filter_chunk_host": (C.c_int, [_vp, C.c_int, C.c_size_t, C.POINTER(C.c_uint), _vp, C.c_size_t,
                                        C.POINTER(_vp), C.POINTER(C.c_size_t)]),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
                "deltarice_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib
