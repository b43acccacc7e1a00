#!/usr/bin/env python3
"""Throughput of the one-chunk host path (what the H5Z callback does): host buffer in, host buffer out,
2000 x 7000 int16 per call.  GB/s of raw int16."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402


def main():
    ctx = dr.Context(0)
    rng = np.random.default_rng(1)
    W, L = 2000, 7000
    x = rng.normal(0, 10, W * L).astype(np.int16)
    opts = (8, L)
    enc = ctx.filter_chunk(x.tobytes(), opts, reverse=False)
    dec = ctx.filter_chunk(enc, opts, reverse=True)
    assert np.array_equal(np.frombuffer(dec, dtype=np.int16), x)
    raw = x.tobytes()
    for name, buf, rev in (("encode", raw, False), ("decode", enc, True)):
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            ctx.filter_chunk(buf, opts, reverse=rev)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[2:]))
        print(f"{name}: {t * 1e3:.3f} ms per 28 MB chunk = {len(raw) / t / 1e9:.2f} GB/s (ratio {len(enc) / len(raw):.4f})", flush=True)


if __name__ == "__main__":
    main()
