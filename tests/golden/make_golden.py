#!/usr/bin/env python3
"""Generates tests/golden/golden.npz from the REFERENCE's own filter.

Run in the container that holds /root/reference (it needs oracle/_ref, built by
`make -C oracle ref` from /root/reference/src/deltaRice.c, unmodified):

    python tests/golden/make_golden.py

Every case is an (opts, words) pair: `words` are the uint32 words the reference's
H5Z_filter_deltarice (src/deltaRice.c:468-490, OpenMP build) emitted for one chunk.
The input chunk is not stored: it is the decode of `words`, and this script asserts
that the reference's decoder returns exactly the generated input before a case is
written (so each case pins encode AND decode).  Cases follow SURVEY.md section 8c.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def gauss(seed, shape, sigma=10.0):
    # README.md:82 / BASELINE config #1: normal(0, sigma).astype(int16) (truncation)
    return np.random.default_rng(seed).normal(0, sigma, shape).astype(np.int16)


def ar1(seed, n_wave, L, rho=0.95, sigma=32.0):
    # marginal sigma, stationary start, rint -> int16
    rng = np.random.default_rng(seed)
    e = rng.normal(0, sigma * np.sqrt(1 - rho * rho), (n_wave, L))
    x = np.empty((n_wave, L))
    x[:, 0] = rng.normal(0, sigma, n_wave)
    for t in range(1, L):
        x[:, t] = rho * x[:, t - 1] + e[:, t]
    return np.rint(x).astype(np.int16)


def from_deltas(d):
    """int16 waveform whose delta-filter residuals are exactly d (mod 2^16)."""
    return np.cumsum(np.asarray(d, dtype=np.int64)).astype(np.uint16).view(np.int16)


def unzig(z):
    z = np.asarray(z, dtype=np.int64)
    return np.where(z & 1, -((z + 1) >> 1), z >> 1)


def cases():
    c = []

    def add(name, x, opts, note):
        c.append((name, np.ascontiguousarray(x), tuple(int(v) for v in opts), note))

    add("kat_docs", np.array([-2, 23], np.int16), (8, 2), "docs/Algorithm.md:9 worked example")
    x1 = gauss(0, (100, 7000))
    add("config1_one_chunk", x1, (8, 7000), "BASELINE config #1, 100x7000 as one chunk")
    for i in range(5):
        add(f"config1_chunk20_{i}", x1[20 * i:20 * i + 20], (8, 7000), "README.md:75-82 chunks=(20,7000)")
    add("zeros_7000", np.zeros(7000, np.int16), (8, 7000), "all-zero waveform: 875 words")
    for k in (1, 2, 3, 4, 15):
        M = 1 << k
        zs = [0, 1, M - 1, M, 8 * M - 2, 8 * M - 1, 8 * M, 8 * M + 1, 65534, 65535, 65535, 0, 8 * M - 1, 8 * M]
        zs = [min(z, 65535) for z in zs]
        add(f"escape_edges_k{k}", from_deltas(unzig(zs)), (M, len(zs)), "codes at the escape boundary z in {8M-1, 8M, 65534, 65535}")
    rng = np.random.default_rng(7)
    u = rng.integers(-32768, 32768, 16384).astype(np.int16)
    add("uniform_default", u, (), "tests/test.py:8-18 shape, default opts (whole chunk one waveform)")
    add("uniform_m16", u, (16,), "tests/test.py:20-31")
    add("uniform_m8_L1024", u, (8, 1024), "tests/test.py:33-44")
    add("uniform_identity", u, (8, 1024, 1, 1), "tests/test.py:46-57 identity prediction filter")
    add("arange_i16_identity", np.arange(-32768, 32768).astype(np.int16), (8, 1024, 1, 1), "tests/test.py:59-70")
    add("arange_u16_identity", np.arange(0, 65536).astype(np.uint16), (8, 1024, 1, 1), "tests/test.py:72-83")
    add("arange_i16_delta", np.arange(-32768, 32768).astype(np.int16), (8, 32768), "examples/testCode.c:32 cd_values")
    add("leftover_20877", gauss(11, 20877), (8, 7000), "trailing partial waveform (6877 samples)")
    g5 = gauss(5, 5000)
    add("cd0", g5, (), "cd_nelmts = 0")
    add("cd1", g5, (4,), "cd_nelmts = 1")
    add("cd2", g5, (4, 500), "cd_nelmts = 2")
    add("cd2_minus1", g5, (8, 0xFFFFFFFF), "WaveformLength = -1 given explicitly")
    g20 = gauss(20, 20000)
    # The reference under-allocates its staging buffer (src/deltaRice.c:411-412 vs :421;
    # SURVEY Appendix B5): it stays inside its own malloc only while
    # n_last <= L - (3W+3)/4, so short waveforms are swept with few of them per chunk.
    for L, W in ((1, 1), (2, 1), (5, 3), (63, 40), (64, 40), (65, 40), (512, 39), (1024, 19),
                 (2048, 9), (7000, 2), (16384, 1)):
        add(f"L{L}", g20[:L * W], (8, L), "waveform-length sweep")
    add("L16384_leftover", g20, (8, 16384), "16384 + leftover 3616")
    add("L_gt_N", g20[:3000], (8, 30000), "WaveformLength > chunk: one short waveform")
    heavy = (np.random.default_rng(9).standard_t(2, 8192) * 40).clip(-32768, 32767).astype(np.int16)
    for k in range(1, 16):
        add(f"k{k}", heavy, (1 << k, 1024), "Rice parameter sweep on heavy-tailed data")
    a = ar1(4321, 20, 7000)
    for m in (4, 8, 16):
        add(f"ar1_m{m}", a, (m, 7000), "BASELINE config #3 shape (AR(1) rho=.95 sigma=32)")
    add("exact_32bits", from_deltas(unzig([0] * 8)), (8, 8), "8 codes x 4 bits = one full word, no padding")
    add("exact_64bits", from_deltas(unzig([0] * 16)), (8, 16), "two full words")
    add("one_sample", np.array([1234], np.int16), (8, 1), "single sample escape (25 bits, one word)")
    g8 = gauss(8, 8192, 30)
    add("fir4", g8, (8, 1024, 4, 1, 0xFFFFFFFF, 1, 0xFFFFFFFF), "4-tap prediction filter [1,-1,1,-1] (docs/Optimization.md:21)")
    add("fir_neg_lead", g8, (8, 1024, 2, 0xFFFFFFFF, 1), "taps[0] = -1")
    return c


def main():
    if not O.have_ref("omp"):
        sys.exit("oracle/_ref missing: run `make -C oracle ref` where /root/reference exists")
    arrays, manifest = {}, []
    for name, x, opts, note in cases():
        xi = x.reshape(-1).view(np.int16)
        words = O.ref_encode_chunk(xi, opts, "omp")
        back = O.ref_decode_chunk(words, opts, "omp")
        assert np.array_equal(back, xi), name
        L = opts[1] if len(opts) >= 2 and opts[1] != 0xFFFFFFFF else xi.size
        if xi.size % L == 0:  # serial build is only well defined without a leftover (Appendix B2)
            assert np.array_equal(O.ref_encode_chunk(xi, opts, "serial"), words), name
        # config #1 re-chunked: inputs are slices of config1_one_chunk, so only the hash is kept
        hash_only = name.startswith("config1_chunk20_")
        if not hash_only:
            arrays[name + "/words"] = words
            arrays[name + "/opts"] = np.array(opts, dtype=np.int64)
        manifest.append({"name": name, "opts": list(opts), "n_samples": int(xi.size),
                         "n_words": int(words.size), "hash_only": hash_only, "sha256_words": hashlib.sha256(words.tobytes()).hexdigest(),
                         "sha256_input": hashlib.sha256(xi.tobytes()).hexdigest(), "note": note})
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
    with open(os.path.join(HERE, "golden_manifest.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "source": "reference src/deltaRice.c compiled unmodified (oracle/Makefile, -fopenmp build)",
                   "cases": manifest}, f, indent=1)
    print(f"{len(manifest)} cases, {os.path.getsize(os.path.join(HERE, 'golden.npz'))} bytes")


if __name__ == "__main__":
    main()
