#!/usr/bin/env python3
"""Rates of the general prediction-filter path (cd_nelmts >= 3) next to the delta path, 200 chunks of 2000 x 7000."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402

ctx = dr.Context(0)
ctx.set_option("profile", 1)
n_chunks, W, L = 200, 2000, 7000
g = torch.Generator(device=ctx.device).manual_seed(7)
x = (torch.randn(n_chunks * W * L, device=ctx.device, generator=g) * 10).to(torch.int16)
torch.cuda.synchronize()
for name, opts in (("delta (default)", (8, L)), ("taps [1,-1] given explicitly", (8, L, 2, 1, 0xFFFFFFFF)),
                   ("taps [1,-1,1,-1]", (8, L, 4, 1, 0xFFFFFFFF, 1, 0xFFFFFFFF)), ("taps [1] (identity)", (8, L, 1, 1))):
    plan = ctx.plan_uniform(n_chunks, W * L, opts)
    words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=ctx.device)
    off = torch.empty(n_chunks + 1, dtype=torch.int64, device=ctx.device)
    y = torch.empty_like(x)
    te, td = [], []
    for _ in range(3):
        plan.encode_async(x, words, off); plan.finish(); te.append(plan.last_timings()[3])
        plan.decode_async(words, off, y); plan.finish(); td.append(plan.last_timings()[3])
    assert torch.equal(x, y)
    b = x.numel() * 2
    print(f"{name:32s} ratio {int(off[-1].item()) * 4 / b:.4f}  encode {b / np.median(te) / 1e6:7.0f} GB/s  decode {b / np.median(td) / 1e6:7.0f} GB/s", flush=True)
    del plan, words, y
