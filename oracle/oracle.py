"""ctypes front end of the parity checker (TEST INFRASTRUCTURE ONLY).

Two checkers live here:

* ``encode_chunk`` / ``decode_chunk`` / ``*_batch`` -- this repo's CPU restatement
  (``oracle/deltarice_oracle.c``; kind "port").
* ``ref_filter`` -- the reference's own ``H5Z_filter_deltarice`` compiled unmodified
  from ``/root/reference/src/deltaRice.c`` into ``oracle/_ref/`` (kind "reference";
  present only when it was built in the container that holds the reference tree;
  the built file travels with the repo snapshot).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  Nothing under ``deltarice_amd/`` does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRO_ORACLE_SO: another build of the same restatement (the sanitizer build of `make -C oracle asan`, tests/test_sanitizers.py)
_ORACLE_SO = os.environ.get("DRO_ORACLE_SO") or os.path.join(_HERE, "libdeltarice_oracle.so")
_REF_SO = {
    "omp": os.path.join(_HERE, "_ref", "libdeltarice_ref_omp.so"),
    "serial": os.path.join(_HERE, "_ref", "libdeltarice_ref_serial.so"),
}

H5Z_FLAG_REVERSE = 0x0100  # H5Zpublic.h


def build(ref: bool = True) -> None:
    """Compiles the restatement and, when the reference tree is present, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", _HERE, "oracle"] + (["ref"] if ref else []), check=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build(ref=False)
        L = C.CDLL(_ORACLE_SO)
        u32p, i16p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_int16), C.POINTER(C.c_uint64)
        L.dro_log2_m.restype = C.c_int
        L.dro_log2_m.argtypes = [C.c_long]
        L.dro_max_chunk_words.restype = C.c_size_t
        L.dro_max_chunk_words.argtypes = [C.c_size_t, C.c_long]
        L.dro_max_wave_words.restype = C.c_size_t
        L.dro_max_wave_words.argtypes = [C.c_size_t]
        L.dro_encode_chunk.restype = C.c_long
        L.dro_encode_chunk.argtypes = [i16p, C.c_size_t, u32p, C.c_size_t, u32p, C.c_size_t]
        for f in (L.dro_decode_chunk, L.dro_decode_chunk_fast):
            f.restype = C.c_long
            f.argtypes = [u32p, C.c_size_t, u32p, C.c_size_t, i16p, C.c_size_t]
        L.dro_rice_pack.restype = C.c_long
        L.dro_rice_pack.argtypes = [i16p, C.c_long, C.c_int, u32p]
        L.dro_rice_unpack.restype = C.c_long
        L.dro_rice_unpack.argtypes = [u32p, C.c_long, C.c_long, C.c_int, i16p]
        L.dro_encode_batch.restype = C.c_long
        L.dro_encode_batch.argtypes = [i16p, C.c_size_t, C.c_size_t, u32p, C.c_size_t, u32p,
                                       C.c_size_t, u64p]
        L.dro_decode_batch.restype = C.c_long
        L.dro_decode_batch.argtypes = [u32p, C.c_size_t, u64p, C.c_size_t, u32p, C.c_size_t, i16p]
        L.dro_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _cd(opts) -> np.ndarray:
    """compression_opts tuple -> cd_values (unsigned); negative ints wrap like HDF5's unsigned."""
    return np.array([int(v) & 0xFFFFFFFF for v in opts], dtype=np.uint32)


def _as_i16(x) -> np.ndarray:
    a = np.ascontiguousarray(x)
    if a.dtype not in (np.int16, np.uint16):
        raise TypeError("chunk data must be 16-bit")
    return a.reshape(-1).view(np.int16)


def max_chunk_words(n_samples: int, wave_len: int = -1) -> int:
    return int(lib().dro_max_chunk_words(n_samples, wave_len))


def encode_chunk(x, opts=()) -> np.ndarray:
    """Encodes one chunk; returns the uint32 words of the filtered chunk."""
    a = _as_i16(x)
    cd = _cd(opts)
    L = int(np.int32(cd[1])) if len(cd) >= 2 else -1
    out = np.empty(max_chunk_words(a.size, L), dtype=np.uint32)
    n = lib().dro_encode_chunk(_p(a, C.c_int16), a.size * 2, _p(cd, C.c_uint32), len(cd),
                               _p(out, C.c_uint32), out.size)
    if n < 0:
        raise ValueError(f"oracle encode failed ({n}) for opts={tuple(opts)}")
    return out[:n].copy()


def decode_chunk(words, opts=(), fast: bool = False) -> np.ndarray:
    w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1)
    cd = _cd(opts)
    if w.size < 2:
        raise ValueError("oracle decode failed: stream too short")
    out = np.empty(int(w[0]), dtype=np.int16)
    f = lib().dro_decode_chunk_fast if fast else lib().dro_decode_chunk
    n = f(_p(w, C.c_uint32), w.size * 4, _p(cd, C.c_uint32), len(cd), _p(out, C.c_int16), out.size)
    if n < 0:
        raise ValueError(f"oracle decode failed ({n}) for opts={tuple(opts)}")
    return out[:n]


def rice_pack(d, k: int) -> np.ndarray:
    a = _as_i16(d)
    out = np.empty(int(lib().dro_max_wave_words(a.size)) + 1, dtype=np.uint32)
    n = lib().dro_rice_pack(_p(a, C.c_int16), a.size, k, _p(out, C.c_uint32))
    return out[:n].copy()


def rice_unpack(words, n: int, k: int):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.empty(n, dtype=np.int16)
    used = lib().dro_rice_unpack(_p(w, C.c_uint32), w.size, n, k, _p(out, C.c_int16))
    return out, int(used)


def encode_batch(x, chunk_samples: int, opts):
    """x: int16 [n_chunks*chunk_samples] -> (words, chunk_word_off[n_chunks+1])."""
    a = _as_i16(x)
    assert a.size % chunk_samples == 0
    nch = a.size // chunk_samples
    cd = _cd(opts)
    L = int(np.int32(cd[1])) if len(cd) >= 2 else -1
    cap = nch * max_chunk_words(chunk_samples, L)
    out = np.empty(cap, dtype=np.uint32)
    off = np.empty(nch + 1, dtype=np.uint64)
    n = lib().dro_encode_batch(_p(a, C.c_int16), nch, chunk_samples, _p(cd, C.c_uint32), len(cd),
                               _p(out, C.c_uint32), cap, _p(off, C.c_uint64))
    if n < 0:
        raise ValueError("oracle batch encode failed")
    return out[:n], off


def decode_batch(words, chunk_word_off, chunk_samples: int, opts) -> np.ndarray:
    w = np.ascontiguousarray(words, dtype=np.uint32)
    off = np.ascontiguousarray(chunk_word_off, dtype=np.uint64)
    nch = off.size - 1
    cd = _cd(opts)
    out = np.empty(nch * chunk_samples, dtype=np.int16)
    n = lib().dro_decode_batch(_p(w, C.c_uint32), nch, _p(off, C.c_uint64), chunk_samples,
                               _p(cd, C.c_uint32), len(cd), _p(out, C.c_int16))
    if n < 0:
        raise ValueError("oracle batch decode failed")
    return out


def num_threads() -> int:
    return int(lib().dro_num_threads())


# ---------------------------------------------------------------------------
# The compiled reference (oracle/_ref)
# ---------------------------------------------------------------------------
_ref = {}
_libc = None


def have_ref(which: str = "omp") -> bool:
    if not os.path.exists(_REF_SO[which]):
        return False
    try:
        _load_ref(which)
        return True
    except OSError:
        return False


def _load_ref(which: str):
    global _libc
    if which not in _ref:
        R = C.CDLL(_REF_SO[which])
        R.H5Z_filter_deltarice.restype = C.c_size_t
        R.H5Z_filter_deltarice.argtypes = [C.c_uint, C.c_size_t, C.POINTER(C.c_uint), C.c_size_t,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        _ref[which] = R
    if _libc is None:
        _libc = C.CDLL(None)
        _libc.malloc.restype = C.c_void_p
        _libc.malloc.argtypes = [C.c_size_t]
        _libc.free.argtypes = [C.c_void_p]
        _libc.realloc.restype = C.c_void_p
        _libc.realloc.argtypes = [C.c_void_p, C.c_size_t]
    return _ref[which]


def ref_filter(data: bytes | np.ndarray, opts=(), reverse: bool = False, which: str = "omp") -> bytes:
    """Runs the reference's H5Z_filter_deltarice (src/deltaRice.c:468-490) on one chunk.

    The callback frees its input and mallocs its output (src/deltaRice.c:433-436,336-340),
    so both buffers go through libc malloc/free."""
    R = _load_ref(which)
    raw = data.tobytes() if isinstance(data, np.ndarray) else bytes(data)
    nbytes = len(raw)
    buf = _libc.malloc(max(nbytes, 16) + 16)
    C.memmove(buf, raw, nbytes)
    # the reference's decoder reads one word past a 1-word waveform
    C.memset(buf + nbytes, 0, 16)
    pbuf = C.c_void_p(buf)
    size = C.c_size_t(nbytes)
    cd = _cd(opts)
    ret = R.H5Z_filter_deltarice(H5Z_FLAG_REVERSE if reverse else 0, len(cd),
                                 _p(cd, C.c_uint) if len(cd) else None, nbytes,
                                 C.byref(size), C.byref(pbuf))
    if ret == C.c_size_t(-1).value:
        _libc.free(pbuf)
        raise ValueError("reference filter returned -1")
    out = C.string_at(pbuf.value, size.value)
    _libc.free(pbuf)
    return out


def ref_encode_chunk(x, opts=(), which="omp") -> np.ndarray:
    return np.frombuffer(ref_filter(_as_i16(x), opts, False, which), dtype=np.uint32).copy()


def ref_decode_chunk(words, opts=(), which="omp") -> np.ndarray:
    w = np.ascontiguousarray(words, dtype=np.uint32)
    return np.frombuffer(ref_filter(w, opts, True, which), dtype=np.int16).copy()
