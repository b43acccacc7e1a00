"""GPU parity across noise levels and RiceParameters: the LDS geometries that follow them.

Round 3's sweeps (profiles/r03_noise_sweep.txt, r03_quiet_sweep.txt) found that every kernel had been tuned on Gaussian
sigma = 10 under m = 8 and that other data took slow paths: the encoders' pieces are sized by the RiceParameter now (drx_internal.h:
pc_seg_samples, fused_wide), the block decoder's geometry by the stream's bits per sample (drx_blocks.hip: blk_segw_bits10).  Each
class of each of them is held to the oracle's bytes here (reference: compressWithRiceCoding / decompressWithRiceCoding,
/root/reference/src/deltaRice.c:191-244,138-189), together with the alternative path a debug flag forces."""
import numpy as np
import pytest

from test_gpu_parity import dev

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

BLOCKS = 4


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


def noise(rng, sigma, n):
    if sigma == 0:
        return np.zeros(n, np.int16)
    return np.clip(rng.normal(0, sigma, n), -32768, 32767).astype(np.int16)


def check(ctx, O, x, n_chunks, N, opts, enc_flags=(0,), dec_flags=(0, 256), want_blocks=None):
    import deltarice_amd as dr
    ref_w, ref_off = O.encode_batch(x, N, opts)
    plan = ctx.plan_uniform(n_chunks, N, opts)
    xd = dev(ctx, x)
    for f in enc_flags:
        ctx.set_option("debug_flags", f)
        w, off = plan.encode(xd).to_numpy()
        ctx.set_option("debug_flags", 0)
        assert np.array_equal(off, ref_off), (opts, f)
        assert np.array_equal(w, ref_w), (opts, f)
    enc = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
    for f in dec_flags:
        ctx.set_option("debug_flags", f)
        y = plan.decode(enc).cpu().numpy()
        path = plan.last_decode_path()
        ctx.set_option("debug_flags", 0)
        assert np.array_equal(y, x), (opts, f)
        if f == 0 and want_blocks is not None:
            assert bool(path & BLOCKS) == want_blocks, (path, opts)
    return ref_w.size * 32.0 / x.size


# (sigma, m): bits per sample from 4 (the reference's default m = 8 on quiet data) to 15
LEVELS = [(0, 8), (1, 8), (3, 8), (10, 8), (10, 64), (40, 32), (80, 64), (160, 128), (320, 256), (1000, 1024), (3000, 2048), (3000, 32768)]


def test_block_decoder_geometry_classes(ctx, O):
    """Few long waveforms: 9 / 11 / 15 / 19 stream words per lane by the stream's bits per sample -- every class, delta and a
    general filter, against the oracle and against the lane-per-waveform decoder."""
    rng = np.random.default_rng(41)
    seen = set()
    for sigma, m in LEVELS:
        x = noise(rng, sigma, 2 * 3 * 90000)
        bits = check(ctx, O, x, 2, 3 * 90000, (m, 90000), want_blocks=True)
        seen.add(9 if bits < 5.4 else 11 if bits < 8.2 else 15 if bits < 11.7 else 19)
    assert seen == {9, 11, 15, 19}, seen
    for sigma, m in [(1, 8), (80, 64), (1000, 1024)]:
        x = noise(rng, sigma, 4 * 70001)
        check(ctx, O, x, 1, 4 * 70001, (m, 70001, 4, 1, 0xFFFFFFFF, 1, 0xFFFFFFFF), want_blocks=True)
    # waveforms of one or two blocks (64 / 128 lanes per block) in the outer classes
    for sigma, m, L in [(1, 8, 5000), (2000, 2048, 4000), (1, 8, 12000), (2000, 2048, 9000)]:
        x = noise(rng, sigma, 3 * 7 * L)
        check(ctx, O, x, 3, 7 * L, (m, L), want_blocks=True)


def test_quiet_stretches_in_noisy_waveforms_take_the_second_parse(ctx, O):
    """Bits per sample are a batch average: a waveform that is quiet for a while puts more codes on a lane than its share of
    the staging buffer holds, and such blocks are decoded in staging passes."""
    rng = np.random.default_rng(42)
    x = noise(rng, 200, 2 * 2 * 120000).reshape(4, 120000)
    x[:, 30000:70000] = 0
    x[1, 90000:] = 7
    x = np.ascontiguousarray(x).reshape(-1)
    check(ctx, O, x, 2, 2 * 120000, (256, 120000), want_blocks=True)


def test_single_pass_encoder_buffer_classes(ctx, O):
    """WaveformLength 7000: 8 waveforms x 2048 LDS words per workgroup while the RiceParameter says a waveform fits, then
    8 x 2496, 4 x 3072, 4 x 4096, then the pieces encoder's segments; flag 65536 = never the larger buffers, 32768 = the pieces
    encoder, 4096 = never the pieces encoder (so that what does not fit is coded twice)."""
    rng = np.random.default_rng(43)
    for sigma, m in LEVELS + [(30000, 32768), (30000, 8)]:
        N = 20 * 7000 - 411  # (a shorter last waveform)
        x = noise(rng, sigma, 3 * N)
        check(ctx, O, x, 3, N, (m, 7000), enc_flags=(0, 65536, 32768, 4096), dec_flags=(0,))
    # a forward filter through the same geometries
    for sigma, m in [(80, 64), (320, 256), (3000, 2048)]:
        x = noise(rng, sigma, 2 * 9 * 8191)
        check(ctx, O, x, 2, 9 * 8191, (m, 8191, 3, 1, 0xFFFFFFFE, 1), enc_flags=(0, 65536, 32768), dec_flags=(0,))


def test_pieces_encoder_sizes_follow_the_rice_parameter(ctx, O):
    """Runs of short waveforms, segments and parts of long ones at every RiceParameter class (pc_seg_samples: 8192 samples per
    segment for m <= 8, 131072 / (2k + 9) beyond), data that fits and data that outgrows the buffer all the same."""
    rng = np.random.default_rng(44)
    shapes = [(2, 40, 1000), (2, 24, 3000), (2, 3, 20000), (1, 2, 70000), (1, 1, 300000), (2, 100, 504)]
    for n_chunks, W, L in shapes:
        for sigma, m in [(10, 8), (80, 64), (1000, 1024), (30000, 32768), (30000, 16), (1, 4096)]:
            N = W * L - L // 3
            x = noise(rng, sigma, n_chunks * N)
            check(ctx, O, x, n_chunks, N, (m, L), enc_flags=(0, 8192), dec_flags=(0,))


def test_streams_of_equal_length_codes(ctx, O):
    """Ramps and square waves code every sample alike ("1010" for slope 1), and a parse from another phase reads the same
    codes: the block decoder's lanes agree with one another on a wrong phase.  The correction is found as a creep (one lane
    per settle round) and applied to all lanes at once, and a block that reads such a pattern does not publish its end before
    its start is verified (drx_blocks.hip: settle()).  Sawtooth wraps put escapes in between; every case must still be the
    oracle's samples, through the block decoder and through the lanes."""
    n = 3 * 300000
    i = np.arange(n)
    cases = {
        "slope 1 sawtooth": (i % 60000 - 30000),
        "slope -1 sawtooth": (30000 - i % 60000),
        "slope 2 sawtooth": (2 * (i % 30000) - 30000),
        "slope 3, short period": (3 * (i % 700) - 1000),
        "square wave": np.where((i // 50) % 2 == 0, 100, -100),
        "alternating +-1": np.where(i % 2 == 0, 7, 8),
        "ramp with noise stretches": np.where((i // 40000) % 3 == 0, np.random.default_rng(46).normal(0, 10, n), i % 60000 - 30000),
    }
    for name, v in cases.items():
        x = np.ascontiguousarray(v).astype(np.int16)
        for L, m in ((300000, 8), (100000, 8), (300000, 64)):
            N = 300000
            check(ctx, O, x, 3, N, (m, L), want_blocks=True)
