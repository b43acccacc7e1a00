#!/usr/bin/env python3
"""Device-resident timing of a single 2000 x 7000 chunk (what one H5Z call runs on the GPU)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402

ctx = dr.Context(0)
ctx.set_option("profile", 1)
W, L = 2000, 7000
x = (torch.randn(W * L, device=ctx.device) * 10).to(torch.int16)
plan = ctx.plan_uniform(1, W * L, (8, L))
words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=ctx.device)
off = torch.empty(2, dtype=torch.int64, device=ctx.device)
y = torch.empty_like(x)
torch.cuda.synchronize()
for impl in (8, 7, 1):
    ctx.set_option("decode_impl", impl)
    for _ in range(3):
        t0 = time.perf_counter(); plan.encode_async(x, words, off); plan.finish(); t1 = time.perf_counter()
        te = plan.last_timings()
        plan.decode_async(words, off, y); plan.finish(); t2 = time.perf_counter()
        td = plan.last_timings()
    print(f"impl {impl}: encode kernel {te[2]:.3f} ms (call {1e3*(t1-t0):.3f}); decode walk {td[0]:.3f} + kernel {td[1]:.3f} ms (call {1e3*(t2-t1):.3f})", flush=True)
assert torch.equal(x, y)
