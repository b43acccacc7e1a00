#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random batch shapes (uniform and ragged), waveform lengths from 1 to
hundreds of thousands, every k, several signal kinds, general filters; GPU encode must equal the oracle's
bytes and GPU decode (default path and the variants a shape can take) must return the input.
usage: python tests/fuzz_parity.py [cases] [seed]   (test infrastructure: it uses the oracle)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402
from oracle import oracle as O  # noqa: E402


def data(rng, kind, n, k):
    if kind == "gauss":
        return rng.normal(0, rng.choice([1, 10, 300, 5000]), n).astype(np.int16)
    if kind == "uniform":
        return rng.integers(-32768, 32768, n).astype(np.int16)
    if kind == "zeros":
        return np.zeros(n, np.int16)
    if kind == "ramp":
        return ((np.arange(n) * int(rng.integers(1, 4))) % 60000 - 30000).astype(np.int16)
    if kind == "pulses":
        x = rng.normal(0, 3, n)
        x += 9000 * ((np.arange(n) % int(rng.integers(50, 5000))) < int(rng.integers(1, 40)))
        return x.astype(np.int16)
    raise ValueError(kind)


def dev(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    scale = int(os.environ.get("DRX_FUZZ_SCALE", 1))  # multiplies the chunk-size caps (bigger, fewer cases)
    ctx = dr.Context(0)
    t0 = time.time()
    lo, hi = 0, cases
    if os.environ.get("DRX_FUZZ_ONLY"):
        lo, hi = (int(v) for v in os.environ["DRX_FUZZ_ONLY"].split(":"))
    logf = open(os.environ["DRX_FUZZ_LOG"], "a") if os.environ.get("DRX_FUZZ_LOG") else None

    def log(msg):
        if logf:
            logf.write(msg + "\n")
            logf.flush()
            os.fsync(logf.fileno())

    for it in range(cases):
        k = int(rng.integers(1, 16)) if rng.random() < 0.9 else 0
        kind = str(rng.choice(["gauss", "uniform", "zeros", "ramp", "pulses"]))
        Lc = [1, 2, 3, 7, 8, 9, 63, 64, 65, 511, 512, 513, 1000, 2047, 2048, 2049, 7000, 8191, 8192, 8193, 16384, 40000,
              65535, 65536, 65537, 100000, 300000]
        ragged = rng.random() < 0.25
        n_chunks = int(rng.integers(1, 6))
        taps = None
        if rng.random() < 0.2:  # (ragged batches too: drx_plan_set_filter applies to every chunk)
            nt = int(rng.integers(1, 6))
            taps = [int(rng.choice([1, -1])) if rng.random() < 0.8 else int(rng.integers(2, 5))] + [int(v) for v in rng.integers(-3, 4, nt - 1)]
        if ragged:
            Ls = [int(rng.choice(Lc)) for _ in range(n_chunks)]
            Ns = [int(min(400000 * scale, max(1, L * int(rng.integers(1, 40 * scale)) + int(rng.integers(0, L))))) for L in Ls]
        else:
            L = int(rng.choice(Lc))
            N = int(min(600000 * scale, max(1, L * int(rng.integers(1, 60 * scale)) + (int(rng.integers(0, L)) if rng.random() < 0.5 else 0))))
            Ls, Ns = [L] * n_chunks, [N] * n_chunks
        if k == 0:
            kind = "zeros" if kind == "uniform" else kind  # M = 1 is defined only while z < 32768
        x = np.concatenate([data(rng, kind, n, k) for n in Ns])
        if k == 0:
            x = (x // 4).astype(np.int16)
        label = f"case {it}: k={k} {kind} chunks={n_chunks} L={Ls} N={Ns} taps={taps}"
        if it < lo or it >= hi:
            continue
        log(label)
        try:
            # oracle stream, chunk by chunk
            words, offs, copts = [], [0], []
            pos = 0
            for n, L in zip(Ns, Ls):
                opts = (1 << k, L) + ((len(taps),) + tuple(t & 0xFFFFFFFF for t in taps) if taps else ())
                w = O.encode_chunk(x[pos:pos + n], opts)
                words.append(w)
                copts.append(opts)
                offs.append(offs[-1] + w.size)
                pos += n
            ref_w = np.concatenate(words)
            ref_off = np.array(offs, np.uint64)
            if ragged:
                plan = ctx.plan(Ns, Ls, 1 << k, taps=taps)
            else:
                opts = (1 << k, Ls[0]) + ((len(taps),) + tuple(t & 0xFFFFFFFF for t in taps) if taps else ())
                plan = ctx.plan_uniform(n_chunks, Ns[0], opts)
            xd = dev(ctx, x)
            # (256: no long-waveform paths; 4096: no pieces encoder; 8192: the segment encoder; 32768: the pieces encoder
            # wherever its geometry allows; 65536: the single-pass encoder's standard geometry only)
            # 4194304: the persistent encoder's segment form wherever the batch is uniform (with 262144: on three workgroups)
            for flags in (0, 256, 4096, 8192, 32768, 65536, 4194304, 4194304 | 262144):
                for eimpl in ((2, 1, 0) if flags in (0, 256) else (2,)):
                    ctx.set_option("encode_impl", eimpl)
                    # (524288: the persistent encoder whatever the batch's size, where the flags leave the choice to it)
                    ctx.set_option("debug_flags", flags | (524288 if eimpl == 2 and flags in (0, 256) else 0))
                    log(f"  encode flags {flags} impl {eimpl}")
                    w, off = plan.encode(xd).to_numpy()
                    assert np.array_equal(off, ref_off), f"offsets (flags {flags}, encoder {eimpl})"
                    assert np.array_equal(w, ref_w), f"stream (flags {flags}, encoder {eimpl})"
            ctx.set_option("encode_impl", 2)
            enc = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
            lossless = taps is None or abs(taps[0]) == 1
            expect = x
            if not lossless:
                expect = np.concatenate([O.decode_chunk(ww, oo) for ww, oo in zip(words, copts)])
            # (16777216: the chunk-wide walk by chains also for these few chunks; 8388608: by reading the chunks)
            for flags, impl in ((0, 8), (256, 8), (512, 8), (0, 7), (256, 7), (0, 0), (131072, 8), (256 | 16777216, 8), (256 | 8388608, 8)):
                ctx.set_option("debug_flags", flags)
                ctx.set_option("decode_impl", impl)
                log(f"  decode flags {flags} impl {impl}")
                y = plan.decode(enc).cpu().numpy()
                assert np.array_equal(y, expect), f"decode (flags {flags}, impl {impl})"
            ctx.set_option("debug_flags", 0)
            ctx.set_option("decode_impl", 8)
            # the encoder's n_i table as a side-band (drx_decode_with_wave_words)
            log("  side-band decode")
            plan.encode(xd)
            table = plan.wave_words_device()
            y = plan.decode_with_wave_words(enc.words, enc.chunk_word_off, table, in_words=enc.total_words)
            plan.finish()
            assert np.array_equal(y.cpu().numpy(), expect), "side-band decode"
            # the one-chunk host path (what the H5Z callback runs) on the first chunk
            opts0 = (1 << k, Ls[0]) + ((len(taps),) + tuple(t & 0xFFFFFFFF for t in taps) if taps else ())
            log("  host path")
            eb = ctx.filter_chunk(x[:Ns[0]], opts0, reverse=False)
            assert eb == words[0].tobytes(), "host path encode"
            db = np.frombuffer(ctx.filter_chunk(eb, opts0, reverse=True), np.int16)
            assert np.array_equal(db, expect[:Ns[0]]), "host path decode"
            if os.environ.get("DRX_FUZZ_CORRUPT"):
                # payload bits flipped (headers intact): any result or DRX_ERR_CORRUPT is fine, a fault is not
                bad = ref_w.copy()
                hdr = set(int(o) for o in ref_off[:-1])
                for _ in range(4):
                    j = int(rng.integers(0, bad.size))
                    if j not in hdr:
                        bad[j] ^= np.uint32(1 << int(rng.integers(0, 32)))
                encb = dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), bad.size)
                for flags, impl in ((0, 8), (256, 8), (512, 8), (0, 7), (0, 0), (256 | 16777216, 8)):
                    ctx.set_option("debug_flags", flags)
                    ctx.set_option("decode_impl", impl)
                    log(f"  corrupt decode flags {flags} impl {impl}")
                    try:
                        plan.decode(encb)
                    except dr.DeltaRiceError:
                        pass
                ctx.set_option("debug_flags", 0)
                ctx.set_option("decode_impl", 8)
        except Exception as e:  # noqa: BLE001
            print("FAIL", label, "->", repr(e), flush=True)
            return 1
        if it % 20 == 19:
            print(f"{it + 1} cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"all {cases} cases ok (seed {seed}, {time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
