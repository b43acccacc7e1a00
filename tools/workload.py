#!/usr/bin/env python3
"""Off-headline workloads, device resident, one JSON line each (HIP events on the codec's stream):

  headline   500 chunks of 2000 x 7000            (bench.py's default; here for A/B under the same harness)
  config5    BASELINE config #5: 100 chunks, WaveformLength in {512, 2048, 7000, 16384}, one ragged batch
  long25     25 chunks of 14 M samples, cd_values = (8): the reference's DEFAULT options, one waveform per chunk
             (src/deltaRice.c:249-258)
  nedm       chunks of 32 x 81 920   (the reference's nEDM@SNS shape, docs/Performance.md:27), 256 chunks
  noptrex    chunks of 32 x 500 000  (NOPTREX, docs/Performance.md:38), 64 chunks
  noptrex_fir4   the same with the prediction filter the reference recommends for NOPTREX, taps [1,-1,1,-1] (docs/Optimization.md:21)
  nedm_fir4      the nEDM shape with that filter
  raglong    a ragged batch of few long waveforms: 48 chunks, WaveformLength in {81 920, 500 000, 250 000, whole chunk}
  raglong_fir4   the same with taps [1,-1,1,-1]
  nab100     100 chunks of 2000 x 7000 (a fifth of the headline batch; with --sigma / --m: noisier data)
  nab1       ONE chunk of 2000 x 7000 (what one H5Z call sees, docs/Performance.md:16)
  small20    ONE chunk of 20 x 7000  (README.md:75-82, BASELINE config #1's chunk)
  small100   ONE chunk of 100 x 7000

usage: workload.py NAME [--steps K] [--m 8] [--no-verify]
The program is meant to be put directly behind `rocprofv3 ... --` (tools/profile_workloads.sh)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402


FIR4 = (1, -1, 1, -1)


def geometry(name):
    """-> (chunk_samples list, wave_len list)"""
    if name.endswith("_fir4"):
        name = name[:-5]
    if name == "raglong":
        Ls, Ns = [], []
        for _ in range(12):
            for L, W in ((81920, 32), (500000, 8), (250000, 16), (0, 1)):
                Ls.append(L)
                Ns.append(L * W if L else 3_000_000)
        return Ns, Ls
    if name == "headline":
        return [2000 * 7000] * 500, [7000] * 500
    if name == "config5":
        Ls, Ns = [], []
        for _ in range(25):
            for L in (512, 2048, 7000, 16384):
                Ls.append(L)
                Ns.append(14_000_000 // L * L)
        return Ns, Ls
    if name == "long25":
        return [14_000_000] * 25, [0] * 25
    if name == "nedm":
        return [32 * 81920] * 256, [81920] * 256
    if name == "noptrex":
        return [32 * 500000] * 64, [500000] * 64
    if name == "nedm_short":  # calibration: the nEDM batch's samples as waveforms of one segment's length (k_encode_stream proper)
        return [32 * 12 * 6832] * 256, [6832] * 256
    if name == "noptrex_short":
        return [32 * 72 * 6952] * 64, [6952] * 64
    if name == "nab100":
        return [2000 * 7000] * 100, [7000] * 100
    if name == "nab1":
        return [2000 * 7000], [7000]
    if name == "small20":
        return [20 * 7000], [7000]
    if name == "small100":
        return [100 * 7000], [7000]
    raise SystemExit(f"unknown workload {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--m", type=int, default=8)
    ap.add_argument("--sigma", type=float, default=10.0, help="standard deviation of the Gaussian samples")
    ap.add_argument("--pulses", action="store_true", help="detector-like data: a quiet baseline (--sigma) with a pulse of amplitude 2000 and "
                    "200 samples of exponential decay every 7000 samples")
    ap.add_argument("--ramp", action="store_true", help="a slope-1 sawtooth instead of noise (a DAQ's test pattern: codes of one length, "
                    "a speculative parse never falls into step)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--encode-only", action="store_true", help="time the encoder alone (ablation builds whose streams do not decode)")
    ap.add_argument("--debug-flags", type=int, default=0)
    ap.add_argument("--sideband", action="store_true", help="decode with the encoder's n_i table as a side-band (drx_decode_with_wave_words)")
    ap.add_argument("--white", action="store_true", help="_fir4 workloads: white noise as it is (the filter then hurts)")
    a = ap.parse_args()
    Ns, Ls = geometry(a.name)
    ctx = dr.Context(0)
    ctx.set_option("profile", 1)
    if a.debug_flags:
        ctx.set_option("debug_flags", a.debug_flags)
    total = sum(Ns)
    g = torch.Generator(device=ctx.device).manual_seed(5)
    x = torch.empty(total, dtype=torch.int16, device=ctx.device)
    slab = 1 << 28
    for s0 in range(0, total, slab):
        n = min(slab, total - s0)
        if a.pulses:
            i = torch.arange(s0, s0 + n, device=ctx.device)
            t = (i % 7000 - 1000).to(torch.float32)
            pulse = torch.where(t >= 0, 2000.0 * torch.exp(-t.clamp(min=0) / 200.0), torch.zeros_like(t))
            x[s0:s0 + n] = (torch.randn(n, device=ctx.device, generator=g) * a.sigma + pulse).to(torch.int16)
        elif a.ramp:
            x[s0:s0 + n] = ((torch.arange(s0, s0 + n, device=ctx.device) % 60000) - 30000).to(torch.int16)
        else:
            x[s0:s0 + n] = (torch.randn(n, device=ctx.device, generator=g) * a.sigma).to(torch.int16)
    uniform = len(set(Ns)) == 1 and len(set(Ls)) == 1
    taps = FIR4 if a.name.endswith("_fir4") else None
    if uniform:
        opts = (a.m, Ls[0] if Ls[0] else Ns[0]) if (Ls[0] or taps) else (a.m,)
        if taps:
            opts = opts + (len(taps),) + tuple(t & 0xFFFFFFFF for t in taps)
        plan = ctx.plan_uniform(len(Ns), Ns[0], opts)
    else:
        plan = ctx.plan(Ns, Ls, a.m, taps=taps)
    words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=ctx.device)
    off = torch.empty(len(Ns) + 1, dtype=torch.int64, device=ctx.device)
    if taps and not a.white:
        # data the filter suits: samples whose RESIDUALS under the filter are the Gaussian noise above (white noise through
        # [1,-1,1,-1] doubles its sigma and escapes 11 % of the samples at m = 8, which is not what the filter is chosen for).
        # Built with the codec itself: the residuals coded with the identity filter are the stream the filter's inverse decodes.
        if uniform:
            ident = ctx.plan_uniform(len(Ns), Ns[0], (a.m, Ls[0] if Ls[0] else Ns[0], 1, 1))
        else:
            ident = ctx.plan(Ns, Ls, a.m, taps=(1,))
        ident.encode_async(x, words, off)
        n0 = ident.finish()
        x = plan.decode_async(words, off, torch.empty_like(x), in_words=n0)
        plan.finish()
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    te, td = [], []
    for i in range(a.steps + 1):
        plan.encode_async(x, words, off)
        nwords = plan.finish()
        t = plan.last_timings()
        if i:
            te.append(t)
        if a.encode_only:
            td.append([0.0, 0.0, 0.0, 1.0])
            continue
        if a.sideband:
            table = plan.wave_words_device()
            plan.decode_with_wave_words(words, off, table, y, in_words=nwords)
        else:
            plan.decode_async(words, off, y, in_words=nwords)
        plan.finish()
        t = plan.last_timings()
        if i:
            td.append(t)
    if not a.no_verify and not a.encode_only:
        assert torch.equal(x, y), "round trip failed"
    te, td = np.median(np.array(te), axis=0), np.median(np.array(td), axis=0)
    raw = total * 2
    ratio = nwords * 4 / raw
    algo = raw * (1 + ratio)
    print(json.dumps({
        "workload": a.name, "sigma": a.sigma, "chunks": len(Ns), "samples": total, "m": a.m, "ratio": ratio, "decode_path": plan.last_decode_path(),
        "encode_ms": {"prepare": float(te[0]), "scan": float(te[1]), "pack": float(te[2]), "total": float(te[3])},
        "decode_ms": {"walk": float(td[0]), "decode": float(td[1]), "total": float(td[3])},
        "encode_GBps_int16": raw / te[3] / 1e6, "decode_GBps_int16": raw / td[3] / 1e6,
        "encode_algorithmic_TBps": algo / te[3] / 1e9, "decode_algorithmic_TBps": algo / td[3] / 1e9,
        "encode_frac_of_8TBps": algo / te[3] / 1e9 / 8.0, "decode_frac_of_8TBps": algo / td[3] / 1e9 / 8.0,
    }), flush=True)


if __name__ == "__main__":
    main()
