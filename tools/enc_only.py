#!/usr/bin/env python3
"""Encoder-only timing of the bench workload under kernel ablation flags (results invalid with flags):
tools/enc_only.py [flags ...]"""
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
    if os.environ.get("DRX_ENCODE_IMPL"):
        ctx.set_option("encode_impl", int(os.environ["DRX_ENCODE_IMPL"]))
    n_chunks, W, L = 500, 2000, 7000
    g = torch.Generator(device=ctx.device).manual_seed(1234)
    x = (torch.randn(n_chunks * W * L, device=ctx.device, generator=g) * 10).to(torch.int16)
    plan = ctx.plan_uniform(n_chunks, W * L, (8, L))
    words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=ctx.device)
    off = torch.empty(n_chunks + 1, dtype=torch.int64, device=ctx.device)
    torch.cuda.synchronize()
    for f in [int(a) for a in sys.argv[1:]] or [0]:
        ctx.set_option("debug_flags", f)
        ts = []
        for _ in range(4):
            plan.encode_async(x, words, off)
            plan.finish()
            ts.append(plan.last_timings()[2])
        print(f"flags {f:4d}  encode kernel {np.median(ts[1:]):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
