"""The H5Z callback from several threads at once ("re-entrant, no global mutable state other than a lazily created
device context guarded by a mutex").  HDF5 itself serialises filter calls of one process, an application with several HDF5
builds / its own threads around the C ABI does not: four threads push chunks of different shapes and options through the
plugin's `H5Z_filter_deltarice` (one shared lazily created context) and through one `drx_ctx` at the same time; every result
must be the reference's bytes."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN = os.path.join(ROOT, "deltarice_amd", "plugin", "libh5deltarice.so")
H5Z_FLAG_REVERSE = 0x100

SHAPES = [  # (samples, opts, kind)
    (20 * 7000, (8, 7000), "gauss"), (100 * 512, (8, 512), "gauss"), (3 * 16384 + 99, (16, 16384), "gauss"), (64 * 300, (8, 64), "gauss"),
    (150000, (8,), "gauss"), (9 * 1000, (4, 1000, 4, 1, 0xFFFFFFFF, 1, 0xFFFFFFFF), "gauss"), (2048 * 6, (8, 2048), "uniform"),
    (40 * 7000, (32, 7000), "gauss"),
]


def _data(rng, n, kind):
    return rng.integers(-32768, 32768, n).astype(np.int16) if kind == "uniform" else rng.normal(0, 25, n).astype(np.int16)


PLUGIN_SCRIPT = r"""
import ctypes as C, sys, threading
import numpy as np
sys.path.insert(0, {root!r})
from oracle import oracle as O
SHAPES = {shapes!r}
p = C.CDLL({plugin!r})
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]
p.H5Z_filter_deltarice.restype = C.c_size_t
p.H5Z_filter_deltarice.argtypes = [C.c_uint, C.c_size_t, C.POINTER(C.c_uint), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
rng = np.random.default_rng(2025)
def data(n, kind):
    return rng.integers(-32768, 32768, n).astype(np.int16) if kind == "uniform" else rng.normal(0, 25, n).astype(np.int16)
cases = [(data(n, kind), opts) for n, opts, kind in SHAPES]
refs = [O.encode_chunk(x, opts).tobytes() for x, opts in cases]
errors = []
def call(flags, opts, payload):
    cd = (C.c_uint * len(opts))(*opts)
    buf = C.c_void_p(libc.malloc(len(payload)))
    C.memmove(buf, payload, len(payload))
    size = C.c_size_t(len(payload))
    ret = p.H5Z_filter_deltarice(flags, len(opts), cd, len(payload), C.byref(size), C.byref(buf))
    out = C.string_at(buf.value, ret) if ret else b""
    libc.free(buf)  # (the callback replaced the buffer; on failure it is still ours)
    return ret, out
def worker(tid):
    try:
        for it in range(12):
            i = (tid * 3 + it) % len(cases)
            x, opts = cases[i]
            ret, enc = call(0, opts, x.tobytes())
            assert ret == len(refs[i]) and enc == refs[i], "thread %d case %d: encode" % (tid, i)
            ret, dec = call(0x100, opts, enc)  # H5Z_FLAG_REVERSE
            assert ret == x.nbytes and dec == x.tobytes(), "thread %d case %d: decode" % (tid, i)
    except Exception as e:
        errors.append(repr(e))
ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
[t.start() for t in ts]
[t.join() for t in ts]
assert not errors, errors[:3]
print("ok")
"""


def test_plugin_callback_from_four_threads():
    # in a child process: the plugin brings the system's HIP runtime, torch (the next test's device memory) its own
    import subprocess
    import sys
    if not os.path.exists(PLUGIN):
        pytest.skip("plugin not built")
    code = PLUGIN_SCRIPT.format(root=ROOT, shapes=SHAPES, plugin=PLUGIN)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


def test_one_context_from_four_threads():
    import deltarice_amd as dr
    from oracle import oracle as O
    ctx = dr.Context(0)
    rng = np.random.default_rng(7)
    cases = [(_data(rng, n, kind), opts) for n, opts, kind in SHAPES]
    refs = [O.encode_chunk(x, opts).tobytes() for x, opts in cases]
    errors = []

    def worker(tid):
        try:
            for it in range(10):
                i = (tid + 2 * it) % len(cases)
                x, opts = cases[i]
                enc = ctx.filter_chunk(x, opts, reverse=False)
                assert enc == refs[i], f"thread {tid} case {i}: encode"
                assert ctx.filter_chunk(enc, opts, reverse=True) == x.tobytes(), f"thread {tid} case {i}: decode"
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    ctx.close()
    assert not errors, errors[:3]


def test_two_contexts_encode_large_batches_at_the_same_time():
    """Two contexts, two streams, two threads, each encoding batches large enough for the persistent encoder
    (`k_encode_stream`: its workgroups stay resident, take tickets and wait for a scanner wavefront of their OWN launch) while the
    other's launch shares the chip: neither may starve the other's scanner or take its tickets, and both streams must be the
    reference's bytes every time."""
    import torch
    import deltarice_amd as dr
    from oracle import oracle as O
    rng = np.random.default_rng(11)
    jobs = []
    for tid, (n_chunks, W, L) in enumerate([(6, 2000, 3000), (5, 2500, 2500)]):
        x = rng.normal(0, 10, n_chunks * W * L).astype(np.int16)
        ref_w, ref_off = O.encode_batch(x, W * L, (8, L))
        jobs.append((n_chunks, W, L, x, ref_w, ref_off))
    errors = []

    def worker(tid):
        try:
            n_chunks, W, L, x, ref_w, ref_off = jobs[tid]
            ctx = dr.Context(0)
            ctx.set_option("debug_flags", 524288)  # the persistent encoder whatever the dispatch would say
            plan = ctx.plan_uniform(n_chunks, W * L, (8, L))
            xd = torch.from_numpy(x).to(ctx.device)
            for it in range(6):
                enc = plan.encode(xd)
                w, off = enc.to_numpy()
                assert np.array_equal(off, ref_off) and np.array_equal(w, ref_w), f"thread {tid} round {it}: encode"
                assert torch.equal(plan.decode(enc), xd), f"thread {tid} round {it}: decode"
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:3]
