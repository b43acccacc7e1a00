"""Direct-chunk HDF5 file <-> VRAM path (include/deltarice_h5io.h, csrc/h5_direct.c): whole datasets move
between a file and HBM with H5Dread_chunk / H5Dwrite_chunk, one PCIe copy and one batched codec call,
instead of one filter callback (two PCIe crossings, one launch) per chunk."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from .codec import Context

H5IO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdeltarice_h5io.so")


class Stats(C.Structure):
    _fields_ = [("rows", C.c_uint64), ("cols", C.c_uint64), ("chunk_rows", C.c_uint64), ("n_chunks", C.c_uint64),
                ("raw_bytes", C.c_uint64), ("stored_bytes", C.c_uint64),
                ("t_file", C.c_double), ("t_pcie", C.c_double), ("t_gpu", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_io = None


def _load():
    global _io
    if _io is None:
        if not os.path.exists(H5IO_PATH):
            raise ImportError(f"{H5IO_PATH} is missing: build it with `make h5io` (needs the HDF5 C library)")
        _lib.load()
        L = C.CDLL(H5IO_PATH)
        L.drx_h5_read.restype = C.c_int
        L.drx_h5_read.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(Stats)]
        L.drx_h5_write.restype = C.c_int
        L.drx_h5_write.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_uint, C.c_uint, C.POINTER(Stats)]
        L.drx_h5_write_filtered.restype = C.c_int
        L.drx_h5_write_filtered.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                            C.c_uint64, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_int32), C.POINTER(Stats)]
        _io = L
    return _io


def read(ctx: Context, path: str, name: str, out: torch.Tensor) -> dict:
    """File -> VRAM.  out: contiguous int16 tensor on ctx.device, large enough for the dataset."""
    st = Stats()
    ctx.stream.wait_stream(torch.cuda.current_stream(ctx.device))
    rc = _load().drx_h5_read(ctx._h, os.fsencode(path), name.encode(), out.data_ptr(), out.numel(), C.byref(st))
    if rc != _lib.DRX_OK:
        raise _lib.DeltaRiceError(rc, f"drx_h5_read({path!r}, {name!r})")
    return st.as_dict()


def write(ctx: Context, path: str, name: str, x: torch.Tensor, rows: int, cols: int, chunk_rows: int,
          rice_m: int = 8, wave_len: int | None = None, taps=None) -> dict:
    """VRAM -> file.  x: contiguous int16 tensor [rows*cols] on ctx.device.  taps: a general prediction filter
    (compression_opts[3:], src/deltaRice.c:277-289 of the reference); None: the delta filter."""
    st = Stats()
    ctx.stream.wait_stream(torch.cuda.current_stream(ctx.device))
    taps = [int(t) for t in (taps or ())]
    tarr = (C.c_int32 * max(len(taps), 1))(*taps)
    rc = _load().drx_h5_write_filtered(ctx._h, os.fsencode(path), name.encode(), x.data_ptr(), rows, cols, chunk_rows,
                                       rice_m, cols if wave_len is None else wave_len, len(taps), tarr, C.byref(st))
    if rc != _lib.DRX_OK:
        raise _lib.DeltaRiceError(rc, f"drx_h5_write({path!r}, {name!r})")
    return st.as_dict()
