#!/usr/bin/env python3
"""Encode/decode rate versus WaveformLength on one MI355X: uniform plans of 14 M-sample chunks, kernel
times from the plan's HIP events (walk / decode, sizes+scan / pack)."""
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
    if os.environ.get("DRX_DEBUG_FLAGS"):
        ctx.set_option("debug_flags", int(os.environ["DRX_DEBUG_FLAGS"]))
    lens = [int(a) for a in sys.argv[1:]] or [64, 512, 2048, 3500, 7000, 16384, 65536]
    print(f"{'L':>7s} {'chunks':>6s} {'enc ms':>8s} {'enc GB/s':>9s} {'walk ms':>8s} {'dec ms':>8s} {'dec GB/s':>9s}")
    for L in lens:
        n_w = 14_000_000 // L
        n_chunks = int(os.environ.get("DRX_SWEEP_CHUNKS", 25))
        N = n_w * L
        g = torch.Generator(device=ctx.device).manual_seed(L)
        x = (torch.randn(n_chunks * N, device=ctx.device, generator=g) * 10).to(torch.int16)
        plan = ctx.plan_uniform(n_chunks, N, (8, L))
        words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=ctx.device)
        off = torch.empty(n_chunks + 1, dtype=torch.int64, device=ctx.device)
        y = torch.empty_like(x)
        torch.cuda.synchronize()  # x was made on torch's default stream, the codec runs on ctx.stream
        te, tw, td, tt = [], [], [], []
        for _ in range(3):
            plan.encode_async(x, words, off)
            if os.environ.get("DRX_SWEEP_ENCODE_ONLY"):  # (ablation builds whose streams do not decode)
                try:
                    plan.finish()
                except dr.DeltaRiceError:
                    pass
                te.append(plan.last_timings()[3]); tw.append(0.0); td.append(0.0); tt.append(1.0)
                continue
            nw = plan.finish()
            te.append(plan.last_timings()[3])
            plan.decode_async(words, off, y, in_words=nw); plan.finish()
            t = plan.last_timings()
            tw.append(t[0]); td.append(t[1]); tt.append(t[3])
        assert os.environ.get("DRX_NO_VERIFY") or torch.equal(x, y)
        b = x.numel() * 2
        print(f"{L:7d} {n_chunks:6d} {np.median(te):8.3f} {b / np.median(te) / 1e6:9.0f} {np.median(tw):8.3f} "
              f"{np.median(td):8.3f} {b / np.median(tt) / 1e6:9.0f}", flush=True)
        del plan, x, y, words


if __name__ == "__main__":
    main()
