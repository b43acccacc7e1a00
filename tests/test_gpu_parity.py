"""GPU parity: the HIP path (through the C ABI) against the oracle and the golden vectors.

Everything here is bit-exact: integer codec, no tolerance."""
import os

import numpy as np
import pytest

from conftest import delta_only, golden_case_names

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx():
    import deltarice_amd as dr
    c = dr.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def dev(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)


def gpu_encode(ctx, plan, x):
    """Encodes with every encoder (the simple size/scan/pack passes; the single pass with a look-back per workgroup and
    its persistent form, both forced with debug flag 256; the pieces encoder wherever its geometry allows, flag 32768; and
    whatever the batch shape selects by default), checks they agree, returns the default one's result."""
    xd = dev(ctx, x.reshape(-1).view(np.int16))
    ctx.set_option("encode_impl", 0)
    enc0 = plan.encode(xd)
    w0, off0 = enc0.to_numpy()
    ctx.set_option("encode_impl", 1)
    ctx.set_option("debug_flags", 256)
    w1, off1 = plan.encode(xd).to_numpy()
    ctx.set_option("encode_impl", 2)
    ctx.set_option("debug_flags", 256 | 524288)  # (524288: the persistent form whatever the batch's size)
    ws, offs = plan.encode(xd).to_numpy()
    assert np.array_equal(offs, off0) and np.array_equal(ws, w0), "persistent single-pass encoder disagrees"
    ctx.set_option("debug_flags", 256 | 4194304)  # (its segment form wherever the batch is uniform)
    wg, offg = plan.encode(xd).to_numpy()
    assert np.array_equal(offg, off0) and np.array_equal(wg, w0), "segment form of the persistent encoder disagrees"
    ctx.set_option("debug_flags", 32768)
    w2, off2 = plan.encode(xd).to_numpy()
    ctx.set_option("debug_flags", 4096)
    w3, off3 = plan.encode(xd).to_numpy()
    ctx.set_option("debug_flags", 0)
    enc = plan.encode(xd)
    w, off = enc.to_numpy()
    assert np.array_equal(off, off0) and np.array_equal(w, w0), "encoder implementations disagree"
    assert np.array_equal(off1, off0) and np.array_equal(w1, w0), "single-pass encoder disagrees"
    assert np.array_equal(off2, off0) and np.array_equal(w2, w0), "pieces encoder disagrees"
    assert np.array_equal(off3, off0) and np.array_equal(w3, w0), "encoder without the pieces path disagrees"
    return enc, w, off


IMPLS = [0, 7, 8]  # every decode_impl the default build offers (include/deltarice_hip.h; 1 / 5 are -DDRX_LEGACY builds')


# --------------------------------------------------------------------------- golden
@pytest.mark.parametrize("name", golden_case_names())
def test_golden_batch_api(ctx, O, golden, name):
    g = golden[name]
    x = O.decode_chunk(g["words"], g["opts"])
    plan = ctx.plan_uniform(1, x.size, g["opts"])
    enc, w, off = gpu_encode(ctx, plan, x)
    assert off.tolist() == [0, g["n_words"]]
    assert np.array_equal(w, g["words"]), "GPU encode differs from the reference's bytes"
    for impl in IMPLS:
        ctx.set_option("decode_impl", impl)
        y = plan.decode(enc).cpu().numpy()
        assert np.array_equal(y, x), f"GPU decode (impl {impl}) differs"
    ctx.set_option("decode_impl", 8)


@pytest.mark.parametrize("name", ["kat_docs", "config1_one_chunk", "leftover_20877", "uniform_default",
                                  "arange_i16_delta", "cd1", "L5", "k15", "uniform_identity",
                                  "arange_i16_identity", "arange_u16_identity", "fir4", "fir_neg_lead"])
def test_golden_filter_callback_semantics(ctx, O, golden, name):
    # the body of H5Z_filter_deltarice: host bytes in, host bytes out
    g = golden[name]
    x = O.decode_chunk(g["words"], g["opts"])
    enc = ctx.filter_chunk(x, g["opts"], reverse=False)
    assert enc == g["words"].tobytes()
    dec = ctx.filter_chunk(g["words"], g["opts"], reverse=True)
    assert dec == x.tobytes()


# --------------------------------------------------------------------------- random vs oracle
def make_data(rng, kind, n):
    if kind == "gauss10":
        return rng.normal(0, 10, n).astype(np.int16)
    if kind == "gauss300":
        return rng.normal(0, 300, n).astype(np.int16)
    if kind == "uniform":
        return rng.integers(-32768, 32768, n).astype(np.int16)
    if kind == "zeros":
        return np.zeros(n, np.int16)
    if kind == "ramp":  # slope-1 ramp: codes of constant length (no self-synchronisation)
        return (np.arange(n) % 60000 - 30000).astype(np.int16)
    if kind == "steps":  # rare huge jumps -> isolated escapes
        x = rng.normal(0, 3, n)
        x[rng.integers(0, n, max(1, n // 500))] += rng.choice([-30000, 30000])
        return x.clip(-32768, 32767).astype(np.int16)
    raise ValueError(kind)


CASES = [
    # (n_chunks, chunk_samples, L, k, kind)
    (3, 14000, 7000, 3, "gauss10"),
    (5, 20 * 7000, 7000, 3, "gauss10"),
    (2, 20877, 7000, 3, "gauss10"),       # leftover waveform
    (4, 4096, 512, 3, "gauss300"),
    (1, 65536, 0, 3, "uniform"),          # whole chunk = one waveform (default opts)
    (2, 10000, 0, 4, "gauss10"),
    (7, 1000, 1, 3, "gauss10"),           # one sample per waveform
    (3, 999, 2, 2, "uniform"),
    (3, 1000, 5, 3, "steps"),
    (2, 6400, 63, 1, "gauss10"),
    (2, 6400, 64, 5, "gauss300"),
    (2, 6500, 65, 15, "uniform"),
    (1, 3000, 30000, 3, "gauss10"),       # L > N
    (9, 2048, 2048, 0, "gauss10"),        # M = 1
    (2, 16384 * 2 + 100, 16384, 3, "gauss10"),
    (65, 513, 513, 3, "gauss10"),         # > 64 waveforms, odd length (unaligned rows)
    (1, 130 * 77, 77, 3, "steps"),        # > 2 decode waves, unaligned rows
    (3, 7000 * 3, 7000, 3, "zeros"),
    (2, 7000 * 4, 7000, 3, "ramp"),
    (2, 7000 * 4, 7000, 1, "ramp"),
    (1, 200000, 100000, 3, "gauss10"),    # long waveforms
    (130, 70, 7, 3, "uniform"),
]


@pytest.mark.parametrize("n_chunks,chunk_samples,L,k,kind", CASES)
def test_random_vs_oracle(ctx, O, n_chunks, chunk_samples, L, k, kind):
    rng = np.random.default_rng(hash((n_chunks, chunk_samples, L, k)) & 0xFFFF)
    x = make_data(rng, kind, n_chunks * chunk_samples)
    opts = (1 << k,) if L == 0 else (1 << k, L)
    ref_w, ref_off = O.encode_batch(x, chunk_samples, opts)
    plan = ctx.plan_uniform(n_chunks, chunk_samples, opts)
    enc, w, off = gpu_encode(ctx, plan, x)
    assert np.array_equal(off, ref_off)
    assert np.array_equal(w, ref_w)
    # n_i table
    nw = plan.wave_words()
    assert int(nw.sum()) + nw.size + n_chunks == ref_w.size
    for impl in IMPLS:
        ctx.set_option("decode_impl", impl)
        y = plan.decode(enc).cpu().numpy()
        assert np.array_equal(y, x), f"impl {impl}"
    ctx.set_option("decode_impl", 8)
    # cross direction: oracle-encoded stream decoded on the GPU
    enc2 = type(enc)(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
    assert np.array_equal(plan.decode(enc2).cpu().numpy(), x)
    # and GPU-encoded stream decoded by the oracle
    assert np.array_equal(O.decode_batch(w, off, chunk_samples, opts), x)


def test_ragged_mixed_waveform_lengths(ctx, O):
    # BASELINE config #5: chunks with WaveformLength in {512, 2048, 7000, 16384}, m = 8, one call
    rng = np.random.default_rng(5)
    Ls = [512, 2048, 7000, 16384, 7000, 512, 16384, 2048, 0]
    Ns = [512 * 40, 2048 * 9 + 17, 7000 * 3, 16384 * 2, 7000 * 2 + 1, 512 * 3, 16384 + 5, 2048, 4321]
    xs = [rng.normal(0, 10, n).astype(np.int16) for n in Ns]
    x = np.concatenate(xs)
    plan = ctx.plan(Ns, Ls, 8)
    enc, w, off = gpu_encode(ctx, plan, x)
    at = 0
    for c, (xc, L) in enumerate(zip(xs, Ls)):
        ref = O.encode_chunk(xc, (8, L) if L else (8,))
        assert off[c] == at
        assert np.array_equal(w[at:at + ref.size], ref), f"chunk {c}"
        at += ref.size
    assert off[-1] == at == enc.total_words
    for impl in IMPLS:
        ctx.set_option("decode_impl", impl)
        assert np.array_equal(plan.decode(enc).cpu().numpy(), x)
    ctx.set_option("decode_impl", 8)


PIECE_CASES = [
    # (n_chunks, chunk_samples, L, k, kind): runs of short waveforms, segments of long ones (drx_pieces.hip)
    (3, 64 * 50 + 7, 64, 3, "gauss10"),        # 16 waveforms per run, partial tiles only, leftover waveform
    (2, 100 * 333, 100, 3, "steps"),
    (4, 512 * 40, 512, 3, "gauss10"),          # whole tiles, 12 per run
    (2, 700 * 29 + 13, 700, 2, "gauss300"),    # a full and a partial tile per waveform
    (3, 1024 * 17, 1024, 3, "gauss10"),
    (2, 2048 * 9 + 17, 2048, 3, "gauss10"),    # three per run
    (2, 3072 * 5, 3072, 3, "gauss10"),         # two per run
    (3, 3073 * 4, 3073, 3, "gauss10"),         # one waveform per wavefront
    (2, 9000 * 3 + 100, 9000, 3, "gauss10"),   # two segments (4608 samples each)
    (2, 16384 * 5, 16384, 3, "gauss10"),
    (2, 20000 * 3 + 4000, 20000, 3, "gauss10"),  # four segments; the leftover waveform fills one of them
    (1, 40000 * 3, 40000, 4, "gauss300"),      # eight segments
    (2, 65536 + 30000, 65536, 3, "gauss10"),
    (2, 512 * 30, 512, 3, "uniform"),          # incompressible: runs outgrow the LDS buffer, coded again to their place
    (2, 2048 * 5, 2048, 3, "uniform"),
    (2, 16384 * 3, 16384, 3, "uniform"),       # ... and so do segments
    (1, 40000 * 2, 40000, 3, "steps"),
    (2, 16384 * 2, 16384, 3, "zeros"),         # 4 bits per sample: segment boundaries inside words
    (2, 512 * 20, 512, 0, "zeros"),            # one bit per sample
    (2, 20000 * 2, 20000, 0, "zeros"),
    (3, 2000 * 10, 2000, 15, "uniform"),
    # packed runs (WaveformLength < 512, a multiple of 8): tiles span waveform boundaries
    (3, 64 * 300 + 13, 64, 3, "gauss10"),          # eight waveforms per tile; a 13-sample leftover waveform
    (2, 8 * 2000 + 3, 8, 3, "gauss10"),            # a waveform per lane
    (2, 16 * 999, 16, 2, "steps"),
    (2, 128 * 77, 128, 3, "gauss300"),
    (3, 200 * 101 + 150, 200, 3, "gauss10"),       # 25 lanes per waveform: boundaries wander through the tiles
    (2, 256 * 40, 256, 4, "gauss10"),
    (2, 504 * 30 + 500, 504, 3, "gauss10"),
    (2, 64 * 200, 64, 3, "uniform"),               # incompressible: the run outgrows the buffer, waveforms coded again one by one
    (2, 128 * 100, 128, 0, "zeros"),               # one bit per sample
    (1, 24 * 700, 24, 15, "uniform"),
    # waveforms over several workgroups (WaveformLength above 65 536): parts of eight segments
    (2, 70000 * 3 + 5000, 70000, 3, "gauss10"),    # two parts; a short leftover waveform
    (3, 81920 * 2, 81920, 3, "gauss10"),           # the reference's nEDM shape (docs/Performance.md:27)
    (1, 200000 * 2 + 65537, 200000, 3, "steps"),   # four parts; leftover of one part + 1 sample
    (2, 300001, 0, 3, "gauss10"),                  # the reference's default: the chunk is one waveform
    (1, 1 << 20, 0, 4, "gauss300"),
    (2, 140000, 140000, 3, "uniform"),             # incompressible: parts coded again, from an unaligned bit on
    (1, 131072 + 40, 131072, 3, "uniform"),        # ... with a second waveform of 40 samples
    (2, 100000, 100000, 0, "zeros"),               # one bit per sample: 32 samples complete a shared word exactly
    (2, 100000 * 2, 100000, 3, "zeros"),
    (1, 65537 * 3, 65537, 3, "ramp"),
    (1, 500000, 250000, 1, "gauss10"),
]


@pytest.mark.parametrize("n_chunks,chunk_samples,L,k,kind", PIECE_CASES)
def test_pieces_encoder_vs_oracle(ctx, O, n_chunks, chunk_samples, L, k, kind):
    rng = np.random.default_rng(hash((n_chunks, chunk_samples, L, k)) & 0xFFFF)
    x = make_data(rng, kind, n_chunks * chunk_samples)
    opts = (1 << k, L) if L else (1 << k,)
    ref_w, ref_off = O.encode_batch(x, chunk_samples, opts)
    plan = ctx.plan_uniform(n_chunks, chunk_samples, opts)
    ctx.set_option("debug_flags", 32768)
    try:
        enc = plan.encode(dev(ctx, x))
        w, off = enc.to_numpy()
        nw = plan.wave_words()
    finally:
        ctx.set_option("debug_flags", 0)
    assert np.array_equal(off, ref_off)
    bad = np.flatnonzero(w != ref_w) if w.size == ref_w.size else None
    assert np.array_equal(w, ref_w), f"first differing word {None if bad is None else bad[:4]}"
    assert int(nw.sum()) + nw.size + n_chunks == ref_w.size
    assert np.array_equal(plan.decode(enc).cpu().numpy(), x)


def test_pieces_encoder_ragged_and_capacity(ctx, O):
    rng = np.random.default_rng(77)
    Ls = [512, 16384, 2048, 7000, 100, 40000, 512, 9000]
    Ns = [512 * 33 + 5, 16384 * 3 + 9000, 2048 * 7, 7000 * 3, 100 * 41, 40000 * 2 + 1, 512, 9000 * 2]
    xs = [make_data(rng, "uniform" if c == 2 else "gauss10", n) for c, n in enumerate(Ns)]
    x = np.concatenate(xs)
    plan = ctx.plan(Ns, Ls, 8)
    ctx.set_option("debug_flags", 32768)
    try:
        enc = plan.encode(dev(ctx, x))
        w, off = enc.to_numpy()
        at = 0
        for c, (xc, L) in enumerate(zip(xs, Ls)):
            ref = O.encode_chunk(xc, (8, L))
            assert off[c] == at
            assert np.array_equal(w[at:at + ref.size], ref), f"chunk {c}"
            at += ref.size
        assert off[-1] == at == enc.total_words
        assert np.array_equal(plan.decode(enc).cpu().numpy(), x)
        # an output buffer one word short: DRX_ERR_CAPACITY
        import deltarice_amd as dr
        for cap in (at - 1, at // 2, 10):
            with pytest.raises(dr.DeltaRiceError) as e:
                plan.encode(dev(ctx, x), capacity_words=cap)
            assert e.value.status == 3
        w2, off2 = plan.encode(dev(ctx, x), capacity_words=at).to_numpy()  # exactly enough; the plan is still usable
        assert np.array_equal(w2, w) and np.array_equal(off2, off)
    finally:
        ctx.set_option("debug_flags", 0)


def test_pieces_encoder_ragged_long_waveforms(ctx, O):
    """Ragged batches whose WaveformLengths are all above 65 536 take the pieces encoder's multi-workgroup form; one that
    mixes them with shorter ones stays with the segment encoder.  Both must give the reference's bytes."""
    rng = np.random.default_rng(78)
    for Ls, Ns in (([70000, 0, 100000, 131072], [70000 * 2 + 9, 250000, 100000 * 3, 131072 + 131071]),
                   ([70000, 512, 100000], [70000 * 2, 512 * 9, 100000]),
                   ([64, 256, 128, 8], [64 * 90 + 5, 256 * 33, 128 * 50, 8 * 700]),     # every chunk packable: packed runs
                   ([64, 7000, 128], [64 * 90, 7000 * 3, 128 * 50])):                  # packable chunks beside others: plain runs
        xs = [make_data(rng, "uniform" if c == 1 else "gauss10", n) for c, n in enumerate(Ns)]
        x = np.concatenate(xs)
        plan = ctx.plan(Ns, Ls, 8)
        enc = plan.encode(dev(ctx, x))
        w, off = enc.to_numpy()
        at = 0
        for c, (xc, L) in enumerate(zip(xs, Ls)):
            ref = O.encode_chunk(xc, (8, L) if L else (8,))
            assert off[c] == at
            assert np.array_equal(w[at:at + ref.size], ref), f"chunk {c}"
            at += ref.size
        assert off[-1] == at == enc.total_words
        assert np.array_equal(plan.decode(enc).cpu().numpy(), x)
        import deltarice_amd as dr
        for cap in (at - 1, at // 3):
            with pytest.raises(dr.DeltaRiceError) as e:
                plan.encode(dev(ctx, x), capacity_words=cap)
            assert e.value.status == 3


def test_decode_chunks_in_arbitrary_order(ctx, O):
    # chunk_word_off for decode is an input: chunks need not be in order or contiguous
    rng = np.random.default_rng(12)
    x = rng.normal(0, 10, (4, 3, 1000)).astype(np.int16)
    opts = (8, 1000)
    streams = [O.encode_chunk(x[c], opts) for c in range(4)]
    order = [2, 0, 3, 1]
    buf, starts, at = [], {}, 3
    buf.append(np.zeros(3, np.uint32))
    for c in order:
        starts[c] = at
        buf.append(streams[c]); buf.append(np.full(5, 0xDEADBEEF, np.uint32))
        at += streams[c].size + 5
    words = np.concatenate(buf)
    plan = ctx.plan_uniform(4, 3000, opts)
    import deltarice_amd as dr
    # offsets table has n+1 entries where entry c+1 is the END of chunk c only for back-to-back
    # chunks; for scattered chunks decode one plan per chunk
    for c in range(4):
        p1 = ctx.plan_uniform(1, 3000, opts)
        off = np.array([starts[c], starts[c] + streams[c].size], np.int64)
        enc = dr.EncodedBatch(dev(ctx, words.view(np.int32)), dev(ctx, off), words.size)
        assert np.array_equal(p1.decode(enc).cpu().numpy(), x[c].ravel())


def test_corrupt_stream_is_rejected_not_crashed(ctx, O):
    import deltarice_amd as dr
    x = np.random.default_rng(1).normal(0, 10, 7000 * 4).astype(np.int16)
    opts = (8, 7000)
    w = O.encode_chunk(x, opts)
    plan = ctx.plan_uniform(1, x.size, opts)
    for mutate in ("n_plus", "n_huge", "total", "truncated"):
        bad = w.copy()
        if mutate == "n_plus":
            bad[1] += 1
        elif mutate == "n_huge":
            bad[1] = 0x7FFFFFFF
        elif mutate == "total":
            bad[0] += 1
        else:
            bad = bad[:-3]
        enc = dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, np.array([0, bad.size], np.int64)), bad.size)
        for impl in IMPLS:
            ctx.set_option("decode_impl", impl)
            with pytest.raises(dr.DeltaRiceError) as e:
                plan.decode(enc)
            assert e.value.status == 4
    ctx.set_option("decode_impl", 8)
    # the one-chunk host path walks the header chain on the CPU: same verdicts
    for bad in (w[:-1], np.concatenate([w[:1] + 1, w[1:]]), np.concatenate([w[:1], w[1:2] + 1, w[2:]]),
                np.concatenate([w[:1], np.array([0x7FFFFFFF], np.uint32), w[2:]]), np.concatenate([w, w[-1:]])):
        with pytest.raises(dr.DeltaRiceError) as e:
            ctx.filter_chunk(bad, opts, reverse=True)
        assert e.value.status == 4
    assert np.array_equal(np.frombuffer(ctx.filter_chunk(w, opts, reverse=True), np.int16), x)


def test_capacity_error(ctx):
    import deltarice_amd as dr
    x = np.random.default_rng(2).integers(-32768, 32768, 7000 * 8).astype(np.int16)
    plan = ctx.plan_uniform(2, 7000 * 4, (8, 7000))
    with pytest.raises(dr.DeltaRiceError) as e:
        plan.encode(dev(ctx, x), capacity_words=1000)
    assert e.value.status == 3
    enc = plan.encode(dev(ctx, x))  # plan still usable
    assert np.array_equal(plan.decode(enc).cpu().numpy(), x)


def test_bad_options_rejected(ctx):
    import deltarice_amd as dr
    for opts in [(0,), (3,), (65536,), (8, 0), (8, 1024, 0), (8, 0x80000000)]:
        with pytest.raises(dr.DeltaRiceError):
            ctx.plan_uniform(1, 1024, opts)
    with pytest.raises(dr.DeltaRiceError):
        ctx.filter_chunk(np.zeros(3, np.uint8), (8, 2))  # odd byte count (src/deltaRice.c:394-397)
    with pytest.raises(dr.DeltaRiceError):
        ctx.filter_chunk(np.zeros(1024, np.int16), (8, 1024, 2, 0, 1))  # taps[0] == 0: the inverse divides by it


# --------------------------------------------------------------------------- size-independent properties
def test_large_batch_properties(ctx, O):
    """BASELINE config #2 shape at 1/50 scale (20000 x 7000, 10 chunks of 2000 x 7000):
    round trip, per-chunk framing invariants, and spot chunks against the oracle."""
    n_chunks, W, L = 10, 2000, 7000
    g = torch.Generator(device=ctx.device).manual_seed(1234)
    x = (torch.randn(n_chunks * W * L, device=ctx.device, generator=g) * 10).to(torch.int16)
    plan = ctx.plan_uniform(n_chunks, W * L, (8, L))
    enc = plan.encode(x)
    y = plan.decode(enc)
    assert torch.equal(x, y)
    ratio = enc.total_words * 4 / (x.numel() * 2)
    assert 0.40 < ratio < 0.41  # 0.4043 for this distribution
    off = enc.chunk_word_off.cpu().numpy()
    nw = plan.wave_words().reshape(n_chunks, W)
    assert np.array_equal(np.diff(off), 1 + W + nw.sum(axis=1))  # checksum of the framing
    hdr = enc.words[torch.from_numpy(off[:-1]).to(ctx.device)].cpu().numpy()
    assert np.all(hdr == W * L)
    for c in (0, 7):
        xc = x[c * W * L:(c + 1) * W * L].cpu().numpy()
        assert enc.chunk_bytes(c) == O.encode_chunk(xc, (8, L)).tobytes()
    # idempotence: encoding the decoded batch gives the same stream
    enc2 = plan.encode(y)
    assert enc2.total_words == enc.total_words
    assert torch.equal(enc2.words[:enc2.total_words], enc.words[:enc.total_words])


def test_full_size_headline_batch(ctx, O):
    """BASELINE config #2 at FULL size (1M x 7000 int16, 500 chunks of 2000 x 7000, m = 8), through what does
    not depend on the size: round trip, the framing's checksum of checksums, header words, spot chunks
    against the oracle bit for bit, idempotence.  About 45 GB of HBM, a few seconds."""
    free, _ = torch.cuda.mem_get_info(ctx.device)
    if free < 60 * 2**30:
        pytest.skip("needs 60 GB of free HBM")
    n_chunks, W, L = 500, 2000, 7000
    x = torch.empty(n_chunks * W * L, dtype=torch.int16, device=ctx.device)
    g = torch.Generator(device=ctx.device).manual_seed(4242)
    slab = 25 * W * L
    for s0 in range(0, x.numel(), slab):
        x[s0:s0 + slab] = torch.randn(slab, device=ctx.device, generator=g).mul_(10.0).to(torch.int16)
    torch.cuda.synchronize()
    plan = ctx.plan_uniform(n_chunks, W * L, (8, L))
    enc = plan.encode(x)
    nw = plan.wave_words().reshape(n_chunks, W)  # n_i as the encoder counted them
    y = plan.decode(enc)
    assert torch.equal(x, y)
    ratio = enc.total_words * 4 / (x.numel() * 2)
    assert 0.4035 < ratio < 0.4050  # 0.4043
    off = enc.chunk_word_off.cpu().numpy()
    assert off[0] == 0 and off[-1] == enc.total_words
    assert np.array_equal(np.diff(off), 1 + W + nw.sum(axis=1, dtype=np.uint64))  # checksum of the framing
    assert np.array_equal(plan.wave_words().reshape(n_chunks, W), nw)               # the decoder's walk found the same n_i
    hdr = enc.words[torch.from_numpy(off[:-1].astype(np.int64)).to(ctx.device)].cpu().numpy()
    assert np.all(hdr == W * L)
    for c in (0, 123, 499):
        xc = x[c * W * L:(c + 1) * W * L].cpu().numpy()
        assert enc.chunk_bytes(c) == O.encode_chunk(xc, (8, L)).tobytes()
    del y
    enc2 = plan.encode(x)
    assert enc2.total_words == enc.total_words
    assert torch.equal(enc2.words[:enc2.total_words], enc.words[:enc.total_words])


@pytest.mark.parametrize("n_chunks,W,L,filt", [(256, 32, 81920, None), (64, 32, 500000, None), (25, 1, 14_000_000, None),
                                               (64, 32, 500000, (1, -1, 1, -1)), (40, 3, 3_000_017, None)])
def test_full_size_long_waveform_batches(ctx, O, n_chunks, W, L, filt):
    """The reference's long-waveform shapes at FULL size (nEDM 256 x 32 x 81 920 and NOPTREX 64 x 32 x 500 000,
    docs/Performance.md:27,38; 25 chunks of one 14 M-sample waveform, its default options; NOPTREX with the filter it
    recommends, docs/Optimization.md:21; a WaveformLength that is no multiple of anything) through the encoder their size
    selects -- the persistent encoder's segment form, k_encode_stream_segs -- and the block decoder: spot chunks against
    the oracle bit for bit, the framing's checksum, round trip, the encoder of round 3 (encode_impl 1) byte for byte."""
    free, _ = torch.cuda.mem_get_info(ctx.device)
    if free < 16 * 2**30:
        pytest.skip("needs 16 GB of free HBM")
    N = W * L
    x = torch.empty(n_chunks * N, dtype=torch.int16, device=ctx.device)
    g = torch.Generator(device=ctx.device).manual_seed(L)
    slab = 1 << 27
    for s0 in range(0, x.numel(), slab):
        n = min(slab, x.numel() - s0)
        x[s0:s0 + n] = torch.randn(n, device=ctx.device, generator=g).mul_(10.0).to(torch.int16)
    torch.cuda.synchronize()
    opts = (8, L) + ((len(filt),) + tuple(t & 0xFFFFFFFF for t in filt) if filt else ())
    plan = ctx.plan_uniform(n_chunks, N, opts)
    ctx.set_option("encode_impl", 2)
    enc = plan.encode(x)
    nw = plan.wave_words().reshape(n_chunks, W)
    off = enc.chunk_word_off.cpu().numpy()
    assert off[0] == 0 and off[-1] == enc.total_words
    assert np.array_equal(np.diff(off), 1 + W + nw.sum(axis=1, dtype=np.uint64))
    for c in sorted({0, n_chunks // 3, n_chunks - 1}):
        xc = x[c * N:(c + 1) * N].cpu().numpy()
        assert enc.chunk_bytes(c) == O.encode_chunk(xc, opts).tobytes(), c
    y = plan.decode(enc)
    assert torch.equal(x, y)
    del y
    ctx.set_option("encode_impl", 1)
    enc1 = plan.encode(x)
    ctx.set_option("encode_impl", 2)
    assert enc1.total_words == enc.total_words
    assert torch.equal(enc1.words[:enc1.total_words], enc.words[:enc.total_words])
    enc2 = plan.encode(x)  # (again, now with the measured bits per sample deciding the segment length)
    assert enc2.total_words == enc.total_words
    assert torch.equal(enc2.words[:enc2.total_words], enc.words[:enc.total_words])


def test_one_huge_chunk_near_the_format_limit(ctx, O):
    """One chunk of 1.5e9 samples (the format allows 2^31 - 1, src/deltaRice.c:389): 32-bit index arithmetic,
    a leftover waveform, 214 286 hops in one header chain.  Round trip, framing, spot waveforms vs the oracle."""
    free, _ = torch.cuda.mem_get_info(ctx.device)
    if free < 24 * 2**30:
        pytest.skip("needs 24 GB of free HBM")
    L, W_full, left = 7000, 214285, 123
    N = W_full * L + left
    g = torch.Generator(device=ctx.device).manual_seed(99)
    x = torch.empty(N, dtype=torch.int16, device=ctx.device)
    slab = 30000 * L
    for s0 in range(0, N, slab):
        n = min(slab, N - s0)
        x[s0:s0 + n] = torch.randn(n, device=ctx.device, generator=g).mul_(10.0).to(torch.int16)
    torch.cuda.synchronize()
    plan = ctx.plan_uniform(1, N, (8, L))
    assert plan.total_waves == W_full + 1
    enc = plan.encode(x)
    nw = plan.wave_words()
    assert torch.equal(plan.decode(enc), x)
    off = enc.chunk_word_off.cpu().numpy()
    assert int(off[1]) == 1 + (W_full + 1) + int(nw.sum(dtype=np.uint64))
    assert int(enc.words[0].item()) == N
    starts = 1 + np.concatenate([[0], np.cumsum(nw.astype(np.uint64) + 1)[:-1]]).astype(np.int64)
    for w in (0, 1, 100000, W_full - 1, W_full):  # the last one is the 123-sample leftover
        xs = x[w * L:min((w + 1) * L, N)].cpu().numpy()
        ref = O.encode_chunk(xs, (8, L))  # one-waveform chunk: [n_samples, n_0, payload...]
        got = enc.words[int(starts[w]):int(starts[w]) + 1 + int(nw[w])].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, ref[1:]), w


def test_general_prediction_filters_vs_oracle(ctx, O):
    """cd_nelmts >= 3 (src/deltaRice.c:64-74,91-102): FIR forward / IIR inverse on the GPU, compared with the
    oracle for taps the reference's own tests and docs use (tests/test.py:46-83, docs/Optimization.md:21)."""
    rng = np.random.default_rng(77)
    x = rng.normal(0, 40, 3 * 5000).astype(np.int16)
    for taps in [(1,), (1, -1, 1, -1), (-1, 1), (1, -2, 1), (2, -1), (1, 0, 0, -1), (-1, 3, -3, 1), (1, 70000, -5),
                 (1, -1, 1, -1, 1)]:
        opts = (8, 1000, len(taps)) + tuple(t & 0xFFFFFFFF for t in taps)
        ref_w, ref_off = O.encode_batch(x, 5000, opts)
        plan = ctx.plan_uniform(3, 5000, opts)
        for eimpl in (0, 1, 2):  # two-pass encoder; single-pass encoders (take up to 4 taps): per workgroup, persistent
            ctx.set_option("encode_impl", eimpl)
            ctx.set_option("debug_flags", 524288 if eimpl == 2 else 0)
            enc = plan.encode(dev(ctx, x))
            w, off = enc.to_numpy()
            assert np.array_equal(off, ref_off) and np.array_equal(w, ref_w), (taps, eimpl)
        ctx.set_option("debug_flags", 0)
        ref_y = O.decode_batch(ref_w, ref_off, 5000, opts)
        for impl in (0, 7, 8):  # simple kernel; staged kernel (taken when taps[0] = +-1 and <= 4 taps), walk separate / fused
            ctx.set_option("decode_impl", impl)
            y = plan.decode(enc).cpu().numpy()
            assert np.array_equal(y, ref_y), (taps, impl)  # lossy taps[0] included
            if abs(taps[0]) == 1:
                assert np.array_equal(y, x), (taps, impl)
        ctx.set_option("decode_impl", 8)


def test_general_filters_through_the_pieces_encoder(ctx, O):
    """Forward filters of up to four taps with short and with long waveforms: runs, segments and multi-workgroup waveforms
    of drx_pieces.hip (the history a filter needs crosses tile, segment and part boundaries, and stops at waveform
    boundaries, src/deltaRice.c:64-74); incompressible data takes the coded-again paths."""
    rng = np.random.default_rng(78)
    for taps in [(1, -1, 1, -1), (-1, 3, -3, 1), (2, -1), (1, -2, 1)]:
        for n_chunks, N, L, kind in [(2, 64 * 150 + 9, 64, "gauss10"), (1, 200 * 60, 200, "gauss300"), (2, 512 * 21 + 100, 512, "gauss10"), (2, 700 * 9, 700, "gauss300"), (1, 16384 * 3 + 999, 16384, "gauss10"),
                                     (2, 70000 * 2 + 30000, 70000, "gauss10"), (1, 150000, 0, "steps"), (1, 2048 * 6, 2048, "uniform"),
                                     (1, 20000 * 2, 20000, "uniform"), (1, 140000, 140000, "uniform")]:
            x = make_data(rng, kind, n_chunks * N)
            opts = (8, L if L else N, len(taps)) + tuple(t & 0xFFFFFFFF for t in taps)
            ref_w, ref_off = O.encode_batch(x, N, opts)
            plan = ctx.plan_uniform(n_chunks, N, opts)
            for flags in (0, 32768, 4096):
                ctx.set_option("debug_flags", flags)
                enc = plan.encode(dev(ctx, x))
                w, off = enc.to_numpy()
                ctx.set_option("debug_flags", 0)
                assert np.array_equal(off, ref_off) and np.array_equal(w, ref_w), (taps, L, kind, flags)
            y = plan.decode(enc).cpu().numpy()
            assert np.array_equal(y, O.decode_batch(ref_w, ref_off, N, opts)), (taps, L, kind)
            if abs(taps[0]) == 1:
                assert np.array_equal(y, x)


def test_persistent_encoder_rings_and_streaming(ctx, O):
    """k_encode_stream (encode_impl 2, round 4): wavefronts on their own, a ring of LDS per wavefront, a scanner workgroup.
    With debug flag 262144 the launch has two coding workgroups, so every wavefront takes many waveforms around its ring:
    code that wraps at the ring's end, waveforms that must wait for the one in front (short behind long), waveforms that
    outgrow the ring (incompressible, or simply long: the streaming path), a shorter last waveform, a general filter, ragged
    chunks, a capacity error -- all held to the oracle's bytes, and to the encoder of round 3."""
    import deltarice_amd as dr
    rng = np.random.default_rng(2024)
    def noisy(n, sigma): return rng.normal(0, sigma, n).astype(np.int16)
    def rough(n): return rng.integers(-32768, 32768, n, dtype=np.int16)
    cases = []
    cases.append(("headline-like", [7000 * 300] * 2, [7000] * 2, 3, None, noisy(7000 * 600, 10)))
    x = noisy(7000 * 200, 10)
    for w in (3, 17, 18, 90, 199):  # incompressible waveforms among compressible ones: 5469 words each, streamed
        x[w * 7000:(w + 1) * 7000] = rough(7000)
    cases.append(("streamed-among-ringed", [7000 * 200], [7000], 3, None, x))
    cases.append(("all-streamed", [7000 * 40 + 123], [7000], 3, None, rough(7000 * 40 + 123)))
    cases.append(("short", [100 * 3000 + 37], [100], 3, None, noisy(100 * 3000 + 37, 10)))
    cases.append(("tiles-exact", [4096 * 150], [4096], 2, None, noisy(4096 * 150, 3)))
    cases.append(("near-ring-size", [9000 * 120], [9000], 4, None, noisy(9000 * 120, 30)))   # ~8.5 bits per sample: ~2400 words
    cases.append(("long", [30000 * 20, 30000 * 7 + 5], [30000, 30000], 3, None, noisy(30000 * 27 + 5, 10)))
    cases.append(("fir4", [5000 * 100], [5000], 3, (1, -1, 1, -1), noisy(5000 * 100, 20)))
    lens = [512, 2048, 7000, 16384, 0, 3333]
    Ns = [512 * 40, 2048 * 9 + 17, 7000 * 30, 16384 * 4, 4321, 3333 * 21 + 1]
    cases.append(("ragged", Ns, lens, 3, None, noisy(sum(Ns), 10)))
    # the segment form (k_encode_stream_segs, flag 4194304: segments of ~1024 samples): long waveforms with a shorter last one,
    # one waveform per chunk, segments that outgrow the ring among ones that do not (the 32 samples coded behind a segment
    # included), a general filter across segment boundaries, a WaveformLength that is no multiple of anything
    cases.append(("segs", [100000 * 5 + 777] * 3, [100000] * 3, 3, None, noisy(3 * (100000 * 5 + 777), 10)))
    cases.append(("segs-whole-chunk", [261001] * 4, [0] * 4, 3, None, noisy(4 * 261001, 25)))
    x = noisy(40013 * 12, 10)
    for at in (0, 5000, 40013 * 3 - 40, 40013 * 7 + 1000, 40013 * 12 - 3000):
        x[at:at + 3000] = rough(3000)
    cases.append(("segs-streamed-among-ringed", [40013 * 12], [40013], 3, None, x))
    cases.append(("segs-fir4", [50000 * 6 + 31] * 2, [50000] * 2, 3, (1, -1, 1, -1), noisy(2 * (50000 * 6 + 31), 20)))
    cases.append(("segs-tiny", [70 * 9 + 5], [70], 3, None, noisy(70 * 9 + 5, 10)))
    for name, Ns, Ls, k, taps, x in cases:
        assert x.size == sum(Ns), name
        uniform = len(set(Ns)) == 1 and len(set(Ls)) == 1
        if uniform:
            opts = ((1 << k, Ls[0]) if Ls[0] else (1 << k,)) + ((len(taps),) + tuple(t & 0xFFFFFFFF for t in taps) if taps else ())
            plan = ctx.plan_uniform(len(Ns), Ns[0], opts)
        else:
            plan = ctx.plan(Ns, Ls, 1 << k)
        words, offs, at = [], [0], 0
        for N, L in zip(Ns, Ls):
            copts = ((1 << k, L) if L else (1 << k,)) + ((len(taps),) + tuple(t & 0xFFFFFFFF for t in taps) if taps else ())
            w = O.encode_chunk(x[at:at + N], copts)
            words.append(w)
            offs.append(offs[-1] + w.size)
            at += N
        ref_w, ref_off = np.concatenate(words), np.array(offs, np.uint64)
        xd = dev(ctx, x)
        for flags in (256 | 4096 | 524288, 256 | 4096 | 524288 | 262144, 256 | 4194304, 256 | 4194304 | 262144):
            ctx.set_option("debug_flags", flags)
            for eimpl in ((2, 1) if flags & 4096 else (2,)):
                ctx.set_option("encode_impl", eimpl)
                enc = plan.encode(xd)
                w, off = enc.to_numpy()
                assert np.array_equal(off, ref_off), (name, flags, eimpl)
                assert np.array_equal(w, ref_w), (name, flags, eimpl)
        # too small an output buffer: reported, nothing written past it
        ctx.set_option("encode_impl", 2)
        cap = int(ref_w.size) - 5
        out = torch.full((cap + 64,), 0x5A5A5A5A, dtype=torch.int32, device=ctx.device)
        off_t = torch.empty(len(Ns) + 1, dtype=torch.int64, device=ctx.device)
        plan.encode_async(xd, out[:cap], off_t)
        with pytest.raises(dr.DeltaRiceError) as ei:
            plan.finish()
        assert ei.value.status == 3, name
        assert bool((out[cap:] == 0x5A5A5A5A).all()), name
        ctx.set_option("debug_flags", 0)
    ctx.set_option("encode_impl", 2)


def test_encoder_dispatch_follows_the_measured_code_length(ctx, O):
    """Which encoder a batch gets (drx_plan_last_encode_path).  The persistent encoder where a waveform's code leaves a ring
    room for the next -- by what the plan's last encode MEASURED, read from the word the encoders write to pinned host memory,
    so also for callers that never wait for an encode and hand the decoder the buffer's capacity as in_words (bench.py does:
    for a while that capacity was taken for the stream's length, and the headline batch went to k_encode_fused); the
    single pass with a look-back per workgroup where the code is long; the segment form for long waveforms."""
    ENC_FUSED, ENC_PIECES, ENC_STREAM, ENC_SEGS = 3, 4, 5, 6
    rng = np.random.default_rng(77)
    W, L, n_chunks = 2100, 7000, 4  # 8400 waveforms: enough for the persistent grid
    for sigma, k, want in ((10, 3, ENC_STREAM), (400, 3, ENC_FUSED)):  # (sigma 400 under m = 8: ~16 bits per sample, 3500 words)
        x = rng.normal(0, sigma, n_chunks * W * L).astype(np.int16)
        xd = dev(ctx, x)
        plan = ctx.plan_uniform(n_chunks, W * L, (1 << k, L))
        words = torch.empty(plan.max_encoded_words, dtype=torch.int32, device=ctx.device)
        off = torch.empty(n_chunks + 1, dtype=torch.int64, device=ctx.device)
        y = torch.empty_like(xd)
        for _ in range(3):  # bench.py's step: no finish() between, in_words = the buffer's capacity
            plan.encode_async(xd, words, off)
            plan.decode_async(words, off, y)
        ctx.stream.synchronize()
        plan.encode_async(xd, words, off)
        assert plan.last_encode_path() == want, (sigma, plan.last_encode_path())
        n = plan.finish()
        ref_w, ref_off = O.encode_batch(x, W * L, (1 << k, L))
        assert n == ref_w.size and np.array_equal(words[:n].cpu().numpy().view(np.uint32), ref_w)
        assert torch.equal(y, xd)
    # long waveforms: the segment form once the batch has 8192 segments, k_encode_pieces below that and with encode_impl 1
    for n_chunks, want in ((2, ENC_PIECES), (40, ENC_SEGS)):
        x = rng.normal(0, 10, n_chunks * 32 * 50000).astype(np.int16)
        plan = ctx.plan_uniform(n_chunks, 32 * 50000, (8, 50000))
        enc = plan.encode(dev(ctx, x))
        assert plan.last_encode_path() == want, (n_chunks, plan.last_encode_path())
        if n_chunks == 2:
            assert enc.chunk_bytes(1) == O.encode_chunk(x[32 * 50000:], (8, 50000)).tobytes()
    ctx.set_option("encode_impl", 1)
    plan.encode(dev(ctx, x))
    assert plan.last_encode_path() == ENC_PIECES
    ctx.set_option("encode_impl", 2)
    # one plan, data of another noise level every time: segments sized for quiet data (7000 samples) meet ~17 bits per sample
    # (every segment outgrows its ring and is coded again to its place), then the measured length shortens them (~2700
    # samples), then quiet data again in those short segments -- the oracle's bytes every time
    N = 32 * 50000
    for i, sigma in enumerate((10, 3000, 3000, 2, 2)):
        x = rng.normal(0, sigma, 40 * N).astype(np.int16)
        enc = plan.encode(dev(ctx, x))
        assert plan.last_encode_path() == ENC_SEGS, (i, sigma)
        for c in (0, 17, 39):
            assert enc.chunk_bytes(c) == O.encode_chunk(x[c * N:(c + 1) * N], (8, 50000)).tobytes(), (i, sigma, c)
        assert np.array_equal(plan.decode(enc).cpu().numpy(), x), (i, sigma)


def test_rice_parameter_optimiser_is_exact(ctx, O):
    rng = np.random.default_rng(31)
    x = (rng.standard_t(3, 4 * 6000) * 25).clip(-32768, 32767).astype(np.int16)
    plan = ctx.plan_uniform(4, 6000, (8, 1500))
    est = plan.estimate_words(dev(ctx, x))
    for k in range(1, 16):
        ref_w, _ = O.encode_batch(x, 6000, (1 << k, 1500))
        assert int(est[k]) == ref_w.size, k
    best = int(np.argmin(est[1:])) + 1
    assert est[best] <= est[3]


def test_filter_and_rice_parameter_optimiser(ctx, O):
    """deltarice_amd.optimise: the neighbourhood search over integer filters of docs/Optimization.md:17-19, M in tandem.  Every
    size it computes is the oracle's encoded size for that filter and RiceParameter; the filter it ends on is a local minimum
    of its (2s+1)^n neighbourhood; on data made for [1,-1,1,-1] it finds that filter from a neighbour of it."""
    from deltarice_amd.optimise import optimise
    rng = np.random.default_rng(17)
    L, W, n_chunks = 3000, 6, 2
    N = W * L
    # (a) a random walk: delta is (near) optimal among two-tap filters
    x = np.cumsum(rng.normal(0, 12, n_chunks * N)).astype(np.int64)
    x = ((x + 32768) % 65536 - 32768).astype(np.int16)
    r = optimise(ctx, dev(ctx, x), N, L, taps=(1, -2), search=1)
    assert r["taps"] == (1, -1), r["taps"]
    for f, words in r["evaluated"].items():
        for k in (0, r["k"], 9):
            opts = (1 << k, L, len(f)) + tuple(t & 0xFFFFFFFF for t in f)
            assert int(words[k]) == O.encode_batch(x, N, opts)[0].size, (f, k)
    best = min(int(w[k]) for w in r["evaluated"].values() for k in range(16))
    assert r["words"] == best  # (nothing it saw is smaller than what it returned)
    for d0 in (-1, 0, 1):  # ... and no valid neighbour of the result was left out
        for d1 in (-1, 0, 1):
            f = (1 + d0, -1 + d1)
            if abs(f[0]) == 1 and f[1] != 0:
                assert f in r["evaluated"], f
    # (b) data whose residuals under [1,-1,1,-1] are small noise: found from a neighbouring filter, with the m that suits
    taps = (1, -1, 1, -1)
    e = rng.normal(0, 6, n_chunks * N).astype(np.int64)
    y = np.zeros(n_chunks * N, np.int64)
    for c in range(n_chunks * W):  # per waveform: y[i] = e[i] + y[i-1] - y[i-2] + y[i-3]  (mod 2^16)
        s = c * L
        for i in range(L):
            acc = e[s + i]
            if i >= 1: acc += y[s + i - 1]
            if i >= 2: acc -= y[s + i - 2]
            if i >= 3: acc += y[s + i - 3]
            y[s + i] = (acc + 32768) % 65536 - 32768
    y = y.astype(np.int16)
    r = optimise(ctx, dev(ctx, y), N, L, taps=(1, -2, 1, -1), search=1)
    assert r["taps"] == taps, r["taps"]
    opts = (r["m"], L, 4) + tuple(t & 0xFFFFFFFF for t in taps)
    assert r["words"] == O.encode_batch(y, N, opts)[0].size
    delta_words = min(int(w) for w in O_sizes(O, y, N, L))
    assert r["words"] < 0.8 * delta_words  # (the point of a filter: well below the best delta coding)


def O_sizes(O, x, N, L):
    return [O.encode_batch(x, N, (1 << k, L))[0].size for k in range(1, 12)]


def test_short_waveform_chunks_walk_through_lds(ctx, O):
    # many short waveforms per chunk: the header chain is walked inside an LDS block (k_walk_block)
    rng = np.random.default_rng(8)
    for L, n_chunks, W in [(512, 3, 700), (100, 2, 5000), (2048, 2, 40), (3, 2, 20000)]:
        x = rng.normal(0, 10, n_chunks * W * L).astype(np.int16)
        opts = (8, L)
        ref_w, ref_off = O.encode_batch(x, W * L, opts)
        plan = ctx.plan_uniform(n_chunks, W * L, opts)
        enc = type(plan.encode(dev(ctx, x)))(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
        for impl in IMPLS:
            ctx.set_option("decode_impl", impl)
            assert np.array_equal(plan.decode(enc).cpu().numpy(), x), (L, impl)
        ctx.set_option("decode_impl", 8)
        bad = ref_w.copy()
        bad[ref_off[1] + 1] += 1  # corrupt the first length header of chunk 1
        import deltarice_amd as dr
        with pytest.raises(dr.DeltaRiceError):
            plan.decode(dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), bad.size))


def test_few_waveforms_take_the_block_decoder(ctx, O):
    """Batches with too few waveforms to give each a lane are decoded a workgroup per block of each waveform's stream
    (drx_blocks.hip): one H5Z call's worth of the README's chunk (20 x 7000), BASELINE config #1's (100 x 7000), an
    nEDM-shaped chunk (32 x 81920), odd lengths, a leftover waveform, waveforms spanning several blocks and several
    staging passes (zeros: 8 samples per word), every code an escape, k from 0 to 15.  Against the oracle's bytes,
    and against the lane-per-waveform decoder (flag 256)."""
    import deltarice_amd as dr
    rng = np.random.default_rng(21)
    shapes = [(1, 20, 7000, 3, "gauss10"), (3, 100, 7000, 3, "gauss10"), (2, 32, 81920, 3, "gauss10"), (1, 7, 4097, 3, "gauss300"),
              (2, 3, 50001, 0, "zeros"), (1, 5, 30011, 3, "zeros"), (1, 4, 65536, 3, "uniform"), (2, 9, 12345, 15, "uniform"),
              (1, 6, 20000, 1, "steps"), (1, 2, 400000, 3, "ramp"), (1, 3, 9000, 7, "gauss300"), (5, 1, 4096, 3, "gauss10")]
    for n_chunks, W, L, k, kind in shapes:
        N = W * L - (L // 3 if W > 2 else 0)  # a shorter last waveform where there is room for one
        x = make_data(rng, kind, n_chunks * N)
        if k == 0:
            x = (x // 4).astype(np.int16)
        opts = (1 << k, L)
        ref_w, ref_off = O.encode_batch(x, N, opts)
        plan = ctx.plan_uniform(n_chunks, N, opts)
        enc = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
        for flags in (0, 256, 512):
            ctx.set_option("debug_flags", flags)
            y = plan.decode(enc).cpu().numpy()
            assert np.array_equal(y, x), (n_chunks, W, L, k, kind, flags)
        ctx.set_option("debug_flags", 0)
        # damage that moves code boundaries is reported, whichever block it hits
        bad = ref_w.copy()
        pos = int(ref_off[0]) + 2 + int(ref_w[int(ref_off[0]) + 1]) // 2
        bad[pos:pos + 3] ^= np.uint32(0x5A5A5A5A)
        encb = dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), bad.size)
        try:
            yb = plan.decode(encb).cpu().numpy()
            assert yb.shape == x.shape  # undetectable damage (remainder bits only) decodes to other values
        except dr.DeltaRiceError as e:
            assert e.status == 4
    assert np.array_equal(plan.decode(enc).cpu().numpy(), x)  # the plan is still usable


def test_few_long_waveforms_take_the_wave_per_waveform_decoder(ctx, O):
    """WaveformLength = -1 (the reference's default: the whole chunk is one waveform) and other long waveforms are
    decoded by a wavefront each with a speculative, self-synchronising parse (k_decode_long).  Noise, a slope-1
    ramp (a mis-started parse of it never re-synchronises: the worst case, 64 restart rounds), full-range
    uniform noise (every code an escape), a constant, and a stream cut short."""
    import deltarice_amd as dr
    rng = np.random.default_rng(12)
    n = 300_017
    cases = {
        "gauss": rng.normal(0, 10, n).astype(np.int16),
        "ramp": (np.arange(n) % 60000 - 30000).astype(np.int16),
        "uniform": rng.integers(-32768, 32768, n).astype(np.int16),
        "zeros": np.zeros(n, np.int16),
        "pulses": (rng.normal(0, 3, n) + 8000 * (np.arange(n) % 5000 < 40)).astype(np.int16),
    }
    for name, x1 in cases.items():
        for opts in ((8,), (1,), (16, 100_000), (8, 65536)):
            x = np.concatenate([x1, x1[::-1], x1])  # three chunks
            L = opts[1] if len(opts) > 1 else n
            ref_w, ref_off = O.encode_batch(x, n, opts)
            plan = ctx.plan_uniform(3, n, (opts[0], L))
            enc = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
            # 256: lane-per-waveform decoder / single-pass encoder; 512: one workgroup per waveform (the block decoder's
            # fallback); 0: the defaults for such batches (a workgroup per block of the stream -- drx_blocks.hip --,
            # a wavefront per 8192-sample segment)
            for flags in (256, 512, 0):
                ctx.set_option("debug_flags", flags)
                assert np.array_equal(plan.decode(enc).cpu().numpy(), x), (name, opts, flags)
                w, off = plan.encode(dev(ctx, x)).to_numpy()
                assert np.array_equal(off, ref_off) and np.array_equal(w, ref_w), (name, opts, flags)
    # regression: zeros at k = 13 never re-synchronise a mis-started parse either; the block-parallel kernel must
    # leave the verdict to the fallback (it once raised DRX_ERR_CORRUPT from its garbage counts)
    z = np.zeros(2 * 600_000, np.int16)
    ref_w, ref_off = O.encode_batch(z, 600_000, (8192, 65536))
    plan = ctx.plan_uniform(2, 600_000, (8192, 65536))
    enc = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
    assert np.array_equal(plan.decode(enc).cpu().numpy(), z)
    # a stream that ends before its waveform does
    x = cases["gauss"]
    w = O.encode_chunk(x, (8,))
    bad = w.copy()
    bad[1] -= 5
    bad = bad[:-5]
    plan = ctx.plan_uniform(1, n, (8, n))
    with pytest.raises(dr.DeltaRiceError):
        plan.decode(dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, np.array([0, bad.size], np.int64)), bad.size))


def test_randomised_shapes_vs_oracle():
    """tests/fuzz_parity.py with a fixed seed: random uniform / ragged batches, WaveformLength from 1 to 300 000,
    every k, five signal kinds, general filters; every encoder and the decoder variants a shape can take."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "120", "11"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_randomised_corrupt_payloads_never_fault():
    """The same generator with DRX_FUZZ_CORRUPT=1: after the parity checks of each case, payload (and waveform
    header) bits are flipped and every decoder variant runs on the damaged stream -- any result or DRX_ERR_CORRUPT
    is acceptable, a fault, a hang or an out-of-bounds access is not (the reference trusts its input completely,
    src/deltaRice.c:138-189,301-358)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DRX_FUZZ_CORRUPT="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "100", "23"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_flipped_payload_bits_that_change_a_code_length_are_reported(ctx, O):
    """A waveform's codes must end inside its last payload word (n_i = ceil(bits / 32), src/deltaRice.c:237-241).  The
    staged and the simple decoders check that after the last sample, so damage that shifts the code boundaries is
    DRX_ERR_CORRUPT instead of silent garbage (damage confined to remainder bits changes values only: the format has
    no checksum)."""
    import deltarice_amd as dr
    rng = np.random.default_rng(3)
    x = rng.normal(0, 10, 64 * 3 * 7000).astype(np.int16)
    opts = (8, 7000)
    w, off = O.encode_batch(x, 64 * 7000, opts)
    plan = ctx.plan_uniform(3, 64 * 7000, opts)
    # all-ones into the middle of a payload: 32 codes of length 4 where ~5 codes were -> the waveform ends early
    pos = int(off[1]) + 1
    for _ in range(17):
        pos += int(w[pos]) + 1  # header of waveform 17 of chunk 1
    bad = w.copy()
    bad[pos + 40:pos + 60] = 0xFFFFFFFF
    enc = dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, off.astype(np.int64)), bad.size)
    for impl in IMPLS:
        ctx.set_option("decode_impl", impl)
        with pytest.raises(dr.DeltaRiceError) as e:
            plan.decode(enc)
        assert e.value.status == 4, impl
    ctx.set_option("decode_impl", 8)
    good = dr.EncodedBatch(dev(ctx, w.view(np.int32)), dev(ctx, off.astype(np.int64)), w.size)
    assert np.array_equal(plan.decode(good).cpu().numpy(), x)
    # a header below the minimum of 1 + k bits per sample (a chunk such as {N, 0, 0, ...} used to decode to zeros)
    z = np.zeros(2 + 875, np.uint32)
    z[0] = 7000
    plan1 = ctx.plan_uniform(1, 7000, opts)
    with pytest.raises(dr.DeltaRiceError) as e:
        plan1.decode(dr.EncodedBatch(dev(ctx, z.view(np.int32)), dev(ctx, np.array([0, z.size], np.int64)), z.size))
    assert e.value.status == 4
    # the host path refuses a header whose sample count the chunk cannot hold, before allocating anything for it
    tiny = np.array([0x7FFFFFF0, 1, 0], np.uint32)
    with pytest.raises(dr.DeltaRiceError) as e:
        ctx.filter_chunk(tiny, opts, reverse=True)
    assert e.value.status == 4


def test_segment_encoder_units_past_a_short_last_waveform(ctx, O):
    """Regression (found by tests/fuzz_parity.py): a chunk of 400 000 samples at WaveformLength 300 000 has a last
    waveform of 100 000 samples; the segment encoder's units past its end once read the sample in front of a
    segment that does not exist -- out of bounds."""
    x = np.zeros(400_000, np.int16)
    x[::7] = 5
    for opts in ((1024, 300_000), (8, 300_000)):
        ref_w, ref_off = O.encode_batch(x, x.size, opts)
        plan = ctx.plan_uniform(1, x.size, opts)
        w, off = plan.encode(dev(ctx, x)).to_numpy()
        assert np.array_equal(off, ref_off) and np.array_equal(w, ref_w)
        assert np.array_equal(plan.decode(plan.encode(dev(ctx, x))).cpu().numpy(), x)


def test_corrupt_headers_through_the_parallel_walks(ctx, O):
    """The parallel header walks (chunk-wide candidates for long waveforms, block-parallel for short ones) must
    reach the same verdict as the serial walkers: a broken chain is DRX_ERR_CORRUPT, never a wrong table."""
    import deltarice_amd as dr
    rng = np.random.default_rng(5)
    for L, W, n_chunks in ((7000, 200, 3), (512, 3000, 2), (3000, 900, 2)):
        x = rng.normal(0, 10, n_chunks * W * L).astype(np.int16)
        opts = (8, L)
        ref_w, ref_off = O.encode_batch(x, W * L, opts)
        plan = ctx.plan_uniform(n_chunks, W * L, opts)
        good = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
        for flags in (0, 16777216, 8388608, 2048):  # parallel walks (chunk-wide: 64 chains chased / the chunk read) / the serial walkers
            ctx.set_option("debug_flags", flags)
            assert np.array_equal(plan.decode(good).cpu().numpy(), x)
        ctx.set_option("debug_flags", 0)
        # header positions of chunk 1
        pos = [int(ref_off[1]) + 1]
        for _ in range(W - 1):
            pos.append(pos[-1] + int(ref_w[pos[-1]]) + 1)
        for which, delta in ((0, 1), (W // 2, 1), (W // 2, -1), (W - 1, 1), (W // 3, 1 << 20), (5, 0x7FFFFFFF)):
            bad = ref_w.copy()
            bad[pos[which]] = (int(bad[pos[which]]) + delta) & 0xFFFFFFFF
            enc = dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), bad.size)
            for flags in (0, 16777216, 8388608, 2048):
                ctx.set_option("debug_flags", flags)
                with pytest.raises(dr.DeltaRiceError) as e:
                    plan.decode(enc)
                assert e.value.status == 4, (L, which, delta, flags)
            ctx.set_option("debug_flags", 0)
        bad = ref_w.copy()
        bad[int(ref_off[1])] += 1  # the chunk's sample count
        with pytest.raises(dr.DeltaRiceError):
            plan.decode(dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), bad.size))


def test_chunk_wide_walk_by_chains(ctx, O):
    """k_walk_sparse: the chunk-wide header walk without reading the chunk -- 64 chains per chunk chased in parallel from starts
    found by looking forward from 64 cuts for a plausible header (validated by the word it points to).  Shapes from few to
    3584 waveforms per chunk, a shorter last waveform, data full of payload words that LOOK like headers (uniform noise under a
    small RiceParameter: escapes whose 16 payload bits end in zeros, followed by eight more), waveforms of very different
    code lengths in one chunk (a chain then collects more headers than its share: the scalar walker takes over), against
    the walk that reads the chunk (flag 8388608) and the serial one (2048)."""
    rng = np.random.default_rng(29)
    cases = []
    for L, W, n_chunks, kind in ((7000, 2000, 6, "gauss"), (2049, 3584, 5, "gauss"), (30000, 64, 6, "gauss"), (7000, 70, 5, "gauss"),
                                 (2100, 5000, 2, "gauss"), (2049, 8192, 1, "gauss"), (7000, 20, 30, "gauss"), (9000, 8, 12, "uniform"),
                                 (5000, 400, 5, "uniform"), (4096, 1000, 5, "uniform"), (7000, 512, 5, "mixed"),
                                 # short waveforms (the block-parallel walk's: the same checks cost nothing here)
                                 (512, 27343, 2, "gauss"), (2048, 6835, 2, "gauss"), (100, 60000, 1, "gauss"), (16, 65536, 1, "gauss"),
                                 (300, 9000, 2, "uniform"), (1000, 1500, 2, "mixed"), (2048, 40, 1, "gauss"), (64, 3000, 1, "zeros")):
        N = W * L - (L // 3 if W > 100 else 0)  # a shorter last waveform
        if kind == "gauss":
            x = rng.normal(0, 10, n_chunks * N).astype(np.int16)
        elif kind == "uniform":
            x = rng.integers(-32768, 32768, n_chunks * N).astype(np.int16)
        elif kind == "zeros":
            x = np.zeros(n_chunks * N, np.int16)
        else:  # the first three quarters of every chunk silent (one bit per sample at k = 0), the rest loud
            x = rng.normal(0, 2000, n_chunks * N).astype(np.int16)
            for c in range(n_chunks):
                x[c * N:c * N + (3 * N) // 4] = 0
        cases.append((L, W, n_chunks, N, kind, x))
    for L, W, n_chunks, N, kind, x in cases:
        k = 0 if kind == "mixed" else 3
        opts = (1 << k, L)
        ref_w, ref_off = O.encode_batch(x, N, opts)
        plan = ctx.plan_uniform(n_chunks, N, opts)
        enc = dr_batch(ctx, ref_w, ref_off)
        # (256: the lane-per-waveform decoder behind the walk, whatever the shape; the other flags: the walks that read the chunks,
        # the serial walkers)
        for flags in (256, 256 | 16777216, 256 | 8388608, 256 | 2048):
            ctx.set_option("debug_flags", flags)
            assert np.array_equal(plan.decode(enc).cpu().numpy(), x), (L, W, kind, flags)
            nw = plan.wave_words()
            assert int(nw.sum(dtype=np.uint64)) + nw.size + n_chunks == ref_w.size, (L, W, kind, flags)
        ctx.set_option("debug_flags", 0)


def dr_batch(ctx, ref_w, ref_off):
    import deltarice_amd as dr
    return dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)


def test_decode_with_the_encoders_table_as_a_side_band(ctx, O):
    """drx_decode_with_wave_words: the n_i table an encode left behind replaces the header walk; the table
    is checked against the stream, a wrong one is DRX_ERR_CORRUPT and never a wild read.  Uniform, ragged (config 5's
    lengths), a general filter, few long waveforms (block decoder behind the side-band)."""
    import deltarice_amd as dr
    rng = np.random.default_rng(91)
    cases = [("uniform", [7000 * 40] * 6, [7000] * 6, None), ("ragged", [512 * 40, 2048 * 9 + 17, 7000 * 3, 16384 * 2, 4321], [512, 2048, 7000, 16384, 0], None),
             ("fir", [3000 * 5 + 11] * 3, [3000] * 3, (1, -1, 1, -1)), ("long", [60000 * 3] * 2, [60000] * 2, None)]
    for name, Ns, Ls, taps in cases:
        x = rng.normal(0, 10, sum(Ns)).astype(np.int16)
        if name == "ragged":
            plan = ctx.plan(Ns, Ls, 8)
        else:
            opts = (8, Ls[0]) + ((len(taps),) + tuple(t & 0xFFFFFFFF for t in taps) if taps else ())
            plan = ctx.plan_uniform(len(Ns), Ns[0], opts)
        enc = plan.encode(dev(ctx, x))
        table = plan.wave_words_device()
        y = plan.decode_with_wave_words(enc.words, enc.chunk_word_off, table, in_words=enc.total_words)
        plan.finish()
        assert np.array_equal(y.cpu().numpy(), x), name
        assert np.array_equal(plan.decode(enc).cpu().numpy(), x)
        # a table that does not belong to the stream
        for damage in ("plus1", "swap", "zero", "huge", "huge_mid"):
            t = table.clone()
            if damage in ("huge", "huge_mid"):
                # one oversized entry in front of valid ones (chunk 0; lane 0 or a lane in the middle of a wavefront): the
                # 32-bit prefix sum of round 3 wrapped behind it and the lanes after it read up to 16 GB in front of the stream
                t[0 if damage == "huge" else min(5, len(t) - 2)] = -16  # 0xFFFFFFF0
            elif damage == "plus1":
                t[len(t) // 2] += 1
            elif damage == "swap" and len(t) > 3 and int(t[0]) != int(t[1]):
                t[0], t[1] = t[1].clone(), t[0].clone()
            elif damage == "zero":
                t[-1] = 0
            else:
                continue
            plan.decode_with_wave_words(enc.words, enc.chunk_word_off, t, in_words=enc.total_words)
            with pytest.raises(dr.DeltaRiceError) as ei:
                plan.finish()
            assert ei.value.status == 4, (name, damage)
        assert np.array_equal(plan.decode(enc).cpu().numpy(), x)  # the plan is still usable


def test_large_chunks_through_the_host_path(ctx, O):
    """One large chunk through host memory (what the H5Z callback sees for the Nab chunk shape, docs/Performance.md:16, and
    larger): the filtered bytes must be the reference's -- odd waveform counts, a shorter last waveform, a general filter,
    16 384-sample waveforms; damaged input is reported.  (Round 3 built a sliced form of this path -- upload, kernels and
    download of waveform slices overlapped through a helper thread -- held it to these cases, and measured no gain: pageable
    copies serialise inside the runtime, profiles/r03_notes.md.)"""
    import deltarice_amd as dr
    rng = np.random.default_rng(404)
    for W, L, short, opts_tail in [(2000, 7000, 0, ()), (1237, 4099, 1234, ()), (700, 16384, 5000, ()), (1500, 3000, 0, (4, 1, 0xFFFFFFFF, 1, 0xFFFFFFFF))]:
        n = W * L - short
        x = rng.normal(0, 10, n).astype(np.int16)
        opts = (8, L) + opts_tail
        ref = O.encode_chunk(x, opts)
        enc = ctx.filter_chunk(x, opts, reverse=False)
        assert enc == ref.tobytes(), (W, L)
        dec = ctx.filter_chunk(ref.tobytes(), opts, reverse=True)
        assert dec == x.tobytes(), (W, L)
        # a broken header chain 60 % into the chunk, a truncated chunk, damaged payload bits
        bad = ref.copy()
        pos, hops = 1, int(W * 0.6)
        for _ in range(hops):
            pos += int(bad[pos]) + 1
        bad[pos] += 3
        with pytest.raises(dr.DeltaRiceError) as ei:
            ctx.filter_chunk(bad.tobytes(), opts, reverse=True)
        assert ei.value.status == 4
        with pytest.raises(dr.DeltaRiceError):
            ctx.filter_chunk(ref[:-5].tobytes(), opts, reverse=True)
        bad = ref.copy()
        bad[pos + 40:pos + 43] ^= np.uint32(0x5A5A5A5A)
        try:
            assert len(ctx.filter_chunk(bad.tobytes(), opts, reverse=True)) == 2 * n
        except dr.DeltaRiceError as e:
            assert e.status == 4
        assert ctx.filter_chunk(ref.tobytes(), opts, reverse=True) == x.tobytes()  # the context is still usable


def test_legacy_decode_variants_are_not_in_the_default_build(ctx):
    import deltarice_amd as dr
    for impl in (1, 5, 2, 9):
        with pytest.raises(dr.DeltaRiceError):
            ctx.set_option("decode_impl", impl)
    ctx.set_option("decode_impl", 8)
