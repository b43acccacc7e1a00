"""Host side of the MI355X Delta-Rice codec: batches of HDF5 chunks resident in HBM.

PyTorch is used only as plumbing (device memory, streams); all arithmetic happens in
the HIP kernels behind the C ABI (include/deltarice_hip.h).  The option tuple has the
meaning of the reference's ``compression_opts`` (README.md:71-80 of the reference,
parsed at src/deltaRice.c:248-291): ``(RiceParameter[, WaveformLength[, nTaps, taps...]])``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import DeltaRiceError, DrxOpts

H5FILTER = 32025  # H5Z_FILTER_DELTARICE (reference src/deltaRice.h:7)


def parse_opts(opts: Sequence[int] = ()) -> DrxOpts:
    """compression_opts -> parsed options; raises DeltaRiceError like the reference rejects
    (stderr + failure, src/deltaRice.c:114-136)."""
    lib = _lib.load()
    cd = (C.c_uint * max(len(opts), 1))(*[int(v) & 0xFFFFFFFF for v in opts])
    o = DrxOpts()
    st = lib.drx_parse_cd_values(len(opts), cd, C.byref(o))
    if st != _lib.DRX_OK:
        raise DeltaRiceError(st, f"compression_opts={tuple(opts)}")
    return o


def _is_delta(o: DrxOpts) -> bool:
    return o.n_taps == 2 and o.taps[0] == 1 and o.taps[1] == -1


@dataclass
class EncodedBatch:
    """Encoded chunks back to back in HBM + the table saying where each one starts."""
    words: torch.Tensor           # int32 storage of the uint32 stream, length = capacity
    chunk_word_off: torch.Tensor  # int64 [n_chunks + 1]
    total_words: int

    def chunk_bytes(self, c: int) -> bytes:
        off = self.chunk_word_off[c:c + 2].cpu().tolist()
        return self.words[off[0]:off[1]].cpu().numpy().view(np.uint32).tobytes()

    def to_numpy(self):
        w = self.words[:self.total_words].cpu().numpy().view(np.uint32)
        return w, self.chunk_word_off.cpu().numpy().astype(np.uint64)


class Context:
    """One GPU, one HIP stream (a torch stream, so torch events/ordering apply to it)."""

    def __init__(self, device: int | torch.device = 0):
        if not torch.cuda.is_available():
            raise DeltaRiceError(2, "no GPU visible: deltarice_amd has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        self.stream = torch.cuda.Stream(device=self.device)
        h = C.c_void_p()
        st = self.lib.drx_ctx_create(self.device.index or 0, C.c_void_p(self.stream.cuda_stream), C.byref(h))
        if st != _lib.DRX_OK:
            raise DeltaRiceError(st, "drx_ctx_create")
        self._h = h
        # A/B timing from the environment (tools/): DRX_ENCODE_IMPL / DRX_DECODE_IMPL = the context options of the same name
        import os
        for key in ("encode_impl", "decode_impl"):
            v = os.environ.get("DRX_" + key.upper())
            if v:
                self.set_option(key, int(v))

    def close(self):
        if getattr(self, "_h", None):
            self.lib.drx_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, st: int):
        if st != _lib.DRX_OK:
            raise DeltaRiceError(st, (self.lib.drx_ctx_last_error(self._h) or b"").decode())

    def set_option(self, key: str, value: int):
        self._check(self.lib.drx_ctx_set_option(self._h, key.encode(), int(value)))

    def synchronize(self):
        self._check(self.lib.drx_ctx_synchronize(self._h))

    def plan_uniform(self, n_chunks: int, chunk_samples: int, opts: Sequence[int] = ()) -> "Plan":
        o = parse_opts(opts)
        h = C.c_void_p()
        L = 0 if o.wave_len < 0 else int(o.wave_len)
        self._check(self.lib.drx_plan_create_uniform(self._h, n_chunks, chunk_samples, L, o.rice_k, C.byref(h)))
        plan = Plan(self, h)
        if not _is_delta(o):  # general prediction filter: GPU FIR/IIR kernels (correct, not tuned)
            self._check(self.lib.drx_plan_set_filter(h, o.n_taps, o.taps))
        return plan

    def plan(self, chunk_samples: Sequence[int], wave_lens: Sequence[int], rice_m: int = 8,
             taps: Optional[Sequence[int]] = None) -> "Plan":
        """Ragged batch: per-chunk sample counts and WaveformLengths (0 or -1: whole chunk); taps: a general
        prediction filter for every chunk (compression_opts[3:] of the reference), None: the delta filter."""
        o = parse_opts((rice_m,))
        n = len(chunk_samples)
        if n == 0 or len(wave_lens) != n:
            raise DeltaRiceError(1, "chunk_samples / wave_lens mismatch")
        cs = (C.c_uint32 * n)(*[int(v) for v in chunk_samples])
        wl = (C.c_uint32 * n)(*[0 if int(v) <= 0 else int(v) for v in wave_lens])
        h = C.c_void_p()
        self._check(self.lib.drx_plan_create(self._h, n, cs, wl, o.rice_k, C.byref(h)))
        plan = Plan(self, h)
        if taps is not None:
            t = (C.c_int32 * len(taps))(*[int(v) for v in taps])
            self._check(self.lib.drx_plan_set_filter(h, len(taps), t))
        return plan

    def filter_chunk(self, data: bytes | np.ndarray, opts: Sequence[int] = (), reverse: bool = False) -> bytes:
        """One chunk through host memory with the H5Z callback's semantics
        (reference src/deltaRice.c:468-490): bytes in, bytes out."""
        raw = data.tobytes() if isinstance(data, np.ndarray) else bytes(data)
        cd = (C.c_uint * max(len(opts), 1))(*[int(v) & 0xFFFFFFFF for v in opts])
        out = C.c_void_p()
        nout = C.c_size_t()
        buf = C.create_string_buffer(raw, len(raw))
        self._check(self.lib.drx_filter_chunk_host(self._h, 1 if reverse else 0, len(opts), cd, buf, len(raw),
                                                   C.byref(out), C.byref(nout)))
        res = C.string_at(out.value, nout.value)
        C.CDLL(None).free(C.c_void_p(out.value))
        return res


class Plan:
    """Geometry of one batch (chunks x waveforms), device resident; reusable."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self._h = ctx, handle
        lib = ctx.lib
        self.n_chunks = int(lib.drx_plan_n_chunks(handle))
        self.total_samples = int(lib.drx_plan_total_samples(handle))
        self.total_waves = int(lib.drx_plan_total_waves(handle))
        self.max_encoded_words = int(lib.drx_plan_max_encoded_words(handle))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx.lib.drx_plan_destroy(self._h)
        self._h = None

    __del__ = close

    def _dev_check(self, t: torch.Tensor, dtype, n: int, name: str):
        if t.device != self.ctx.device or t.dtype != dtype or not t.is_contiguous() or t.numel() < n:
            raise DeltaRiceError(1, f"{name}: need contiguous {dtype} tensor of >= {n} elements on {self.ctx.device}")

    def encode_async(self, x: torch.Tensor, out_words: Optional[torch.Tensor] = None,
                     chunk_word_off: Optional[torch.Tensor] = None, capacity_words: Optional[int] = None):
        """Launches the encode on the context's stream; no host sync."""
        self._dev_check(x, torch.int16, self.total_samples, "x")
        if out_words is None:
            cap = self.max_encoded_words if capacity_words is None else int(capacity_words)
            out_words = torch.empty(cap, dtype=torch.int32, device=self.ctx.device)
        if chunk_word_off is None:
            chunk_word_off = torch.empty(self.n_chunks + 1, dtype=torch.int64, device=self.ctx.device)
        self._dev_check(chunk_word_off, torch.int64, self.n_chunks + 1, "chunk_word_off")
        self.ctx._check(self.ctx.lib.drx_encode(self._h, x.data_ptr(), out_words.data_ptr(), out_words.numel(),
                                                chunk_word_off.data_ptr()))
        return out_words, chunk_word_off

    def finish(self) -> int:
        """Waits for the last call on this plan and raises on device-side errors
        (capacity, corrupt input); returns the encoded word count of the last encode."""
        n = C.c_uint64()
        self.ctx._check(self.ctx.lib.drx_plan_finish(self._h, C.byref(n)))
        return int(n.value)

    def encode(self, x: torch.Tensor, capacity_words: Optional[int] = None) -> EncodedBatch:
        cur = torch.cuda.current_stream(self.ctx.device)
        self.ctx.stream.wait_stream(cur)
        words, off = self.encode_async(x, capacity_words=capacity_words)
        total = self.finish()
        return EncodedBatch(words, off, total)

    def decode_async(self, words: torch.Tensor, chunk_word_off: torch.Tensor, out: Optional[torch.Tensor] = None,
                     in_words: Optional[int] = None):
        self._dev_check(words, torch.int32, 1, "words")
        self._dev_check(chunk_word_off, torch.int64, self.n_chunks + 1, "chunk_word_off")
        if out is None:
            out = torch.empty(self.total_samples, dtype=torch.int16, device=self.ctx.device)
        self._dev_check(out, torch.int16, self.total_samples, "out")
        n = words.numel() if in_words is None else int(in_words)
        self.ctx._check(self.ctx.lib.drx_decode(self._h, words.data_ptr(), n, chunk_word_off.data_ptr(),
                                                out.data_ptr()))
        return out

    def decode_with_wave_words(self, words: torch.Tensor, chunk_word_off: torch.Tensor, wave_words: torch.Tensor,
                               out: Optional[torch.Tensor] = None, in_words: Optional[int] = None) -> torch.Tensor:
        """Decode with the encoder's n_i table as a side-band (int32 tensor [total_waves] on the device): no header walk.
        Launch only; finish() raises DRX_ERR_CORRUPT if the table does not belong to the stream."""
        self._dev_check(words, torch.int32, 1, "words")
        self._dev_check(chunk_word_off, torch.int64, self.n_chunks + 1, "chunk_word_off")
        self._dev_check(wave_words, torch.int32, self.total_waves, "wave_words")
        if out is None:
            out = torch.empty(self.total_samples, dtype=torch.int16, device=self.ctx.device)
        self._dev_check(out, torch.int16, self.total_samples, "out")
        n = words.numel() if in_words is None else int(in_words)
        self.ctx._check(self.ctx.lib.drx_decode_with_wave_words(self._h, words.data_ptr(), n, chunk_word_off.data_ptr(),
                                                                wave_words.data_ptr(), out.data_ptr()))
        return out

    def wave_words_device(self) -> torch.Tensor:
        """A device copy of n_i of every waveform from the last encode/decode (the side-band of decode_with_wave_words)."""
        return torch.from_numpy(self.wave_words().view(np.int32)).to(self.ctx.device)

    def decode(self, enc: EncodedBatch, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        cur = torch.cuda.current_stream(self.ctx.device)
        self.ctx.stream.wait_stream(cur)
        y = self.decode_async(enc.words, enc.chunk_word_off, out, in_words=enc.total_words)
        self.finish()
        return y

    def estimate_words(self, x: torch.Tensor) -> np.ndarray:
        """Exact encoded size (uint32 words) of this batch for RiceParameter 2^k, k = 0..15 -- the
        optimisation the reference's docs/Optimization.md describes; argmin gives the best m."""
        self._dev_check(x, torch.int16, self.total_samples, "x")
        cur = torch.cuda.current_stream(self.ctx.device)
        self.ctx.stream.wait_stream(cur)
        out = (C.c_uint64 * 16)()
        self.ctx._check(self.ctx.lib.drx_estimate_words(self._h, x.data_ptr(), out))
        return np.array(list(out), dtype=np.uint64)

    def last_timings(self):
        """Kernel times (ms) of the last call, HIP events on the context's stream; needs
        ctx.set_option("profile", 1).  encode: (sizes, scan, pack, total); decode: (walk, decode, 0, total)."""
        ms = (C.c_float * 4)()
        self.ctx._check(self.ctx.lib.drx_plan_last_timings(self._h, ms))
        return tuple(float(v) for v in ms)

    def last_decode_path(self) -> int:
        """DRX_PATH_* bits (include/deltarice_hip.h) of the decoders the last decode used."""
        return int(self.ctx.lib.drx_plan_last_decode_path(self._h))

    def last_encode_path(self) -> int:
        """DRX_ENC_* (include/deltarice_hip.h): the encoder the last encode used."""
        return int(self.ctx.lib.drx_plan_last_encode_path(self._h))

    def wave_words(self) -> np.ndarray:
        """n_i (payload words) of every waveform from the last encode/decode, on the host."""
        buf = np.empty(self.total_waves, dtype=np.uint32)
        self.ctx._check(self.ctx.lib.drx_plan_read_wave_words(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint32))))
        return buf
