#!/usr/bin/env python3
"""Is the headline decoder's run-to-run spread (5.0 ... 5.4 ms) a matter of WHERE its buffers lie?  One process, one encoded
batch; the decode's output (and, second table, its input) is moved through one large allocation in steps, three decodes each."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402


def main():
    ctx = dr.Context(0)
    ctx.set_option("profile", 1)
    n_chunks, W, L = 500, 2000, 7000
    N = n_chunks * W * L
    g = torch.Generator(device=ctx.device).manual_seed(1)
    x = torch.empty(N, dtype=torch.int16, device=ctx.device)
    slab = 25 * W * L
    for s0 in range(0, N, slab):
        x[s0:s0 + slab] = torch.randn(slab, device=ctx.device, generator=g).mul_(10.0).to(torch.int16)
    torch.cuda.synchronize()
    plan = ctx.plan_uniform(n_chunks, W * L, (8, L))
    enc = plan.encode(x)
    nw = enc.total_words
    pad = 1 << 22  # int16 elements of slack (8 MB)
    ybig = torch.empty(N + pad, dtype=torch.int16, device=ctx.device)
    wbig = torch.empty(nw + pad, dtype=torch.int32, device=ctx.device)

    def run(words, y):
        t = []
        for _ in range(4):
            plan.decode_async(words, enc.chunk_word_off, y, in_words=nw)
            plan.finish()
            t.append(plan.last_timings()[1])
        return min(t[1:]), float(np.median(t[1:]))

    print("output offset (bytes)   decode ms (min, median of 3)   [data_ptr mod 2 MB]")
    for off in (0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, (1 << 21) + 4096):
        y = ybig[off // 2: off // 2 + N]
        lo, med = run(enc.words, y)
        print(f"{off:12d}   {lo:7.3f} {med:7.3f}   {y.data_ptr() % (1 << 21)}", flush=True)
    print("input offset (bytes)")
    y = ybig[:N]
    for off in (0, 64, 256, 1024, 4096, 16384, 65536, 1 << 20):
        w = wbig[off // 4: off // 4 + nw]
        w.copy_(enc.words[:nw])
        torch.cuda.synchronize()
        lo, med = run(w, y)
        print(f"{off:12d}   {lo:7.3f} {med:7.3f}   {w.data_ptr() % (1 << 21)}", flush=True)
    print("the same buffers again, five times (spread inside one process):")
    for _ in range(5):
        print("   %.3f %.3f" % run(enc.words, ybig[:N]), flush=True)


if __name__ == "__main__":
    main()
