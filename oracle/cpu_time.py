#!/usr/bin/env python3
"""CPU baseline leg of bench.py (TEST INFRASTRUCTURE: this is the checker timed, never the product).

Runs in a child process so that the OpenMP team size is fixed by the environment
(OMP_NUM_THREADS) before libgomp starts -- the reference publishes a 1-thread and an
all-threads figure (docs/Performance.md:24-25) and so does bench.py.

  python3 oracle/cpu_time.py chunks.npy M L budget_seconds

chunks.npy: int16 [n_chunks, samples_per_chunk].  Times ONLY the filter call
(H5Z_filter_deltarice of src/deltaRice.c when oracle/_ref is present, else this repo's
restatement): the malloc'ed input is filled before the timer starts and the result is read
after it stops (src/deltaRice.c:468-490 takes and returns malloc'ed buffers).  Prints one JSON line.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path[0] = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # the script dir would shadow the package
from oracle import oracle as O  # noqa: E402


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def time_reference(chunks, opts, budget):
    R = O._load_ref("omp")
    libc = O._libc
    cd = O._cd(opts)
    cdp = O._p(cd, C.c_uint)
    t_enc = t_dec = 0.0
    raw = done = 0
    t_start = time.perf_counter()
    for xc in chunks:
        nbytes = xc.nbytes
        buf = libc.malloc(nbytes + 16)
        C.memmove(buf, xc.ctypes.data, nbytes)  # outside the timer
        pbuf, size = C.c_void_p(buf), C.c_size_t(nbytes)
        t0 = time.perf_counter()
        ret = R.H5Z_filter_deltarice(0, len(cd), cdp, nbytes, C.byref(size), C.byref(pbuf))
        t1 = time.perf_counter()
        assert ret != C.c_size_t(-1).value, "reference encode failed"
        enc_bytes = size.value
        # the callback frees its input and returns a malloc'ed buffer: feed that buffer back, grown by 16 zero
        # bytes outside the timer (the reference's decoder may read one word past the stream)
        pbuf = C.c_void_p(libc.realloc(pbuf, enc_bytes + 16))
        C.memset(pbuf.value + enc_bytes, 0, 16)
        size2 = C.c_size_t(enc_bytes)
        t2 = time.perf_counter()
        ret = R.H5Z_filter_deltarice(O.H5Z_FLAG_REVERSE, len(cd), cdp, enc_bytes, C.byref(size2), C.byref(pbuf))
        t3 = time.perf_counter()
        assert ret != C.c_size_t(-1).value and size2.value == nbytes, "reference decode failed"
        ok = C.string_at(pbuf.value, nbytes) == xc.tobytes()  # after the timer
        libc.free(pbuf)
        assert ok, "cpu baseline round trip failed"
        t_enc += t1 - t0
        t_dec += t3 - t2
        raw += nbytes
        done += 1
        if time.perf_counter() - t_start > budget:
            break
    return t_enc, t_dec, raw, done


def time_port(chunks, opts, budget):
    L = O.lib()
    cd = O._cd(opts)
    wl = int(np.int32(cd[1])) if len(cd) >= 2 else -1
    t_enc = t_dec = 0.0
    raw = done = 0
    t_start = time.perf_counter()
    for xc in chunks:
        a = np.ascontiguousarray(xc).reshape(-1)
        out = np.empty(O.max_chunk_words(a.size, wl), dtype=np.uint32)
        y = np.empty(a.size, dtype=np.int16)
        t0 = time.perf_counter()
        n = L.dro_encode_chunk(O._p(a, C.c_int16), a.size * 2, O._p(cd, C.c_uint32), len(cd), O._p(out, C.c_uint32), out.size)
        t1 = time.perf_counter()
        assert n > 0
        t2 = time.perf_counter()
        m = L.dro_decode_chunk_fast(O._p(out, C.c_uint32), n * 4, O._p(cd, C.c_uint32), len(cd), O._p(y, C.c_int16), y.size)
        t3 = time.perf_counter()
        assert m == a.size and np.array_equal(y, a), "cpu baseline round trip failed"
        t_enc += t1 - t0
        t_dec += t3 - t2
        raw += a.nbytes
        done += 1
        if time.perf_counter() - t_start > budget:
            break
    return t_enc, t_dec, raw, done


def main():
    path, m, wl, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    chunks = np.load(path, mmap_mode="r")
    chunks = [np.ascontiguousarray(chunks[i]) for i in range(chunks.shape[0])]
    opts = (m, wl)
    use_ref = O.have_ref("omp")
    threads = O.num_threads()
    # one untimed call: page faults of first touch, the OpenMP team's start-up
    (time_reference if use_ref else time_port)(chunks[:1], opts, 1e9)
    t_enc, t_dec, raw, done = (time_reference if use_ref else time_port)(chunks, opts, budget)
    print(json.dumps({
        "kind": "reference" if use_ref else "port", "threads": threads, "cpu": cpu_model(),
        "chunks": done, "raw_bytes": raw, "t_enc": t_enc, "t_dec": t_dec,
        "encode_GBps": raw / t_enc / 1e9, "decode_GBps": raw / t_dec / 1e9,
        "value": raw / (t_enc + t_dec) / 1e9,
    }))


if __name__ == "__main__":
    main()
