"""CPU: pins the oracle (oracle/deltarice_oracle.c) to the reference.

Golden vectors are bytes emitted by the reference's own compiled filter
(tests/golden/make_golden.py); the docs' worked example is checked literally."""
import hashlib

import numpy as np
import pytest

from conftest import golden_case_names
from oracle import oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_docs_worked_example():
    # /root/reference/docs/Algorithm.md:9: x=-2 -> 1011, x'=50 -> 0000001010 at m=8
    w = O.encode_chunk(np.array([-2, 23], np.int16), (8, 2))
    assert w.tolist() == [2, 1, 0xB0280000]
    assert O.decode_chunk(w, (8, 2)).tolist() == [-2, 23]


@pytest.mark.parametrize("name", golden_case_names())
def test_golden_decode_then_encode(golden, name):
    g = golden[name]
    x = O.decode_chunk(g["words"], g["opts"])
    assert x.size == g["n_samples"]
    assert sha(x) == g["sha256_input"], "oracle decode differs from the reference's input"
    assert np.array_equal(O.decode_chunk(g["words"], g["opts"], fast=True), x)
    w = O.encode_chunk(x, g["opts"])
    assert np.array_equal(w, g["words"]), "oracle encode differs from the reference's bytes"
    assert sha(w) == g["sha256_words"]


def test_config1_rechunked(golden):
    # README.md:75-82: 100x7000 written with chunks=(20,7000): five filter calls
    x = O.decode_chunk(golden["config1_one_chunk"]["words"], (8, 7000)).reshape(100, 7000)
    for i in range(5):
        g = golden[f"config1_chunk20_{i}"]
        w = O.encode_chunk(x[20 * i:20 * i + 20], g["opts"])
        assert sha(w) == g["sha256_words"] and w.size == g["n_words"]
    assert golden["config1_one_chunk"]["n_words"] * 4 == 566488


def test_ratio_anchor_config1(golden):
    g = golden["config1_one_chunk"]
    assert abs(g["n_words"] * 4 / (g["n_samples"] * 2) - 0.40463) < 1e-5


def test_zero_waveform_is_875_words(golden):
    w = golden["zeros_7000"]["words"]
    assert w[0] == 7000 and w[1] == 875 and w.size == 877
    assert np.all(w[2:] == 0x88888888)


def test_rice_primitives_roundtrip():
    rng = np.random.default_rng(3)
    for k in range(0, 16):
        d = (rng.standard_t(2, 3000) * (1 << k) / 2).clip(-32768, 32767).astype(np.int16)
        d[:4] = [-32768, 32767, 0, -1]
        w = O.rice_pack(d, k)
        back, used = O.rice_unpack(w, d.size, k)
        assert np.array_equal(back, d)
        assert (used + 31) // 32 == w.size


def test_rejects_bad_options():
    x = np.zeros(64, np.int16)
    for opts in [(0,), (3,), (65536,), (8, 0), (8, 1024, 0), (8, 1024, 2, 0, 1), (8, 0x80000000)]:
        with pytest.raises(ValueError):
            O.encode_chunk(x, opts)


def test_decoder_rejects_corrupt_chain():
    w = O.encode_chunk(np.arange(100, dtype=np.int16), (8, 10))
    bad = w.copy()
    bad[1] += 1
    with pytest.raises(ValueError):
        O.decode_chunk(bad, (8, 10))
    with pytest.raises(ValueError):
        O.decode_chunk(w[:-1], (8, 10))


@pytest.mark.skipif(not O.have_ref("omp"), reason="oracle/_ref not built (needs /root/reference)")
def test_against_compiled_reference_random():
    rng = np.random.default_rng(2024)
    for trial in range(40):
        n = int(rng.integers(200, 9000))
        k = int(rng.integers(1, 16))
        L = int(rng.integers(150, n + 50))
        sigma = float(rng.choice([1, 5, 30, 300, 6000]))
        x = rng.normal(0, sigma, n).clip(-32768, 32767).astype(np.int16)
        opts = (1 << k, L)
        # stay inside the reference's own allocation
        W = -(-n // L)
        if W > 1 and L < 4 * W:
            continue
        w = O.encode_chunk(x, opts)
        assert np.array_equal(O.ref_encode_chunk(x, opts), w), (n, k, L, sigma)
        assert np.array_equal(O.ref_decode_chunk(w, opts), x)
