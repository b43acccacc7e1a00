#!/usr/bin/env python3
"""BASELINE.json configs 3 and 5 on one MI355X (config 2 is bench.py's default):
  3: 1M x 7000 AR(1) rho=0.95 sigma=32, m in {4, 8, 16}
  5: mixed WaveformLength {512, 2048, 7000, 16384} chunks, m = 8, one batch call (ragged plan)
Each line: ratio, encode / decode GB/s of int16, round trip verified bit-exact on the GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402


def run_bench(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--no-collect", "--steps", "3",
                          "--warmup", "1"] + extra, capture_output=True, text=True, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def timed(ctx, fn, reps=3):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ctx.stream); fn(); b.record(ctx.stream); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    rows = []
    for m in (4, 8, 16):
        d = run_bench(["--dist", "ar1", "--m", str(m)])
        rows.append((f"config 3: 1M x 7000 AR(1), m={m}", d["compression_ratio"], d["encode_GBps"], d["decode_GBps"]))
    # config 5: the same number of samples per WaveformLength as one Nab chunk set (14M samples each), 25 chunk sets
    ctx = dr.Context(0)
    Ls, Ns = [], []
    for rep in range(25):
        for L in (512, 2048, 7000, 16384):
            n_w = 14_000_000 // L
            Ls.append(L); Ns.append(n_w * L)
    total = sum(Ns)
    g = torch.Generator(device=ctx.device).manual_seed(5)
    x = (torch.randn(total, device=ctx.device, generator=g) * 10).to(torch.int16)
    plan = ctx.plan(Ns, Ls, 8)
    words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=ctx.device)
    off = torch.empty(len(Ns) + 1, dtype=torch.int64, device=ctx.device)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    t_enc = timed(ctx, lambda: plan.encode_async(x, words, off))
    plan.finish()
    nwords = int(off[-1].item())
    t_dec = timed(ctx, lambda: plan.decode_async(words, off, y))
    plan.finish()
    assert torch.equal(x, y)
    rows.append((f"config 5: {len(Ns)} chunks, L in {{512,2048,7000,16384}}, {total / 1e9:.2f} G samples, m=8",
                 nwords * 4 / (total * 2), total * 2 / t_enc / 1e6, total * 2 / t_dec / 1e6))
    print(f"{'workload':78s} {'ratio':>7s} {'enc GB/s':>9s} {'dec GB/s':>9s}")
    for r in rows:
        print(f"{r[0]:78s} {r[1]:7.4f} {r[2]:9.0f} {r[3]:9.0f}")


if __name__ == "__main__":
    main()
