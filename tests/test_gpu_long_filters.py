"""GPU parity: batches of FEW LONG waveforms with a general prediction filter or a ragged geometry -- what the reference
recommends for NOPTREX (32 x 500 000 samples per chunk, taps [1,-1,1,-1]: /root/reference/docs/Optimization.md:21,
docs/Performance.md:38, src/deltaRice.c:91-102).  They take the block-parallel decoder (drx_blocks.hip); a general filter's
inverse then runs in place over the residuals, parallel inside a waveform (drx_iir.hip).  Every case is held to the oracle's
bytes and to the lane-per-waveform decoder (debug flag 256), and the path a decode took is read back from the plan."""
import numpy as np
import pytest

from test_gpu_parity import dev, make_data

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

BLOCKS, IIR, IIR_FUSED, LANES_ANY = 4, 32, 64, 1 | 2


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


TAPS = [(1, -1, 1, -1), (-1, 1), (1, -2, 1), (1, 3, -5, 7), (-1, -1, 2, 5), (1,), (-1,), (1, 0, 0, -1)]


def u32(taps):
    return tuple(t & 0xFFFFFFFF for t in taps)


def test_general_filter_few_long_waveforms_uniform(ctx, O):
    import deltarice_amd as dr
    rng = np.random.default_rng(301)
    shapes = [(2, 5, 60000, 3, "gauss10"), (1, 32, 81920, 3, "gauss10"), (2, 4, 30011, 8, "gauss300"), (1, 6, 9000, 3, "uniform"),
              (3, 1, 150000, 3, "steps"), (1, 3, 8193, 3, "gauss10"), (1, 2, 8192, 3, "gauss10"), (1, 3, 24577, 3, "gauss10"),
              (1, 7, 4097, 15, "uniform"), (1, 2, 2500000, 3, "gauss10"),  # (77 tiles per waveform)
              # whole tiles (32 768 samples: moved through LDS) at an odd sample offset, a waveform of whole tiles only, a
              # last tile that ends inside its first / behind its seventh wavefront
              (1, 3, 40001, 3, "gauss10"), (2, 2, 65536, 3, "gauss10"), (1, 2, 32768 + 100, 3, "gauss10"), (1, 2, 32768 + 7 * 4096 + 5, 3, "gauss300")]
    for si, (n_chunks, W, L, k, kind) in enumerate(shapes):
        N = W * L - (L // 3 if W > 2 else 0)  # a shorter last waveform where there is room for one
        x = make_data(rng, kind, n_chunks * N)
        for taps in (TAPS if si < 3 else TAPS[:2]):
            opts = (1 << k, L, len(taps)) + u32(taps)
            ref_w, ref_off = O.encode_batch(x, N, opts)
            plan = ctx.plan_uniform(n_chunks, N, opts)
            enc = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
            for flags in (0, 256):
                ctx.set_option("debug_flags", flags)
                y = plan.decode(enc).cpu().numpy()
                path = plan.last_decode_path()
                ctx.set_option("debug_flags", 0)
                assert np.array_equal(y, x), (n_chunks, W, L, k, kind, taps, flags)
                if flags == 0:
                    assert path & BLOCKS and path & IIR, (path, W, L)
                else:
                    assert path & LANES_ANY and not path & BLOCKS, (path, W, L)
            # ... and what the GPU encodes decodes to the same samples
            enc2 = plan.encode(dev(ctx, x))
            w2, off2 = enc2.to_numpy()
            assert np.array_equal(off2, ref_off) and np.array_equal(w2, ref_w)


def test_general_filter_many_long_waveforms_fused(ctx, O):
    """Batches with at least as many long waveforms as the block decoder keeps workgroups resident (NOPTREX-shaped: 64 chunks
    of 32 x 500 000; nEDM: 256 x 32 x 81 920): the inverse filter runs INSIDE the block decoder (round 4: one kernel, samples
    straight to the output; DRX_PATH_IIR_FUSED), its state handed from block to block and from run to run of a waveform.
    Held to the oracle's stream, to the two-pass form (flag 2097152) and to the lane-per-waveform decoder (flag 256): one, two
    and three blocks per waveform, runs of two blocks, odd WaveformLengths (the staging buffer's first sample then sits
    anywhere in its 16-byte piece), a shorter last waveform, quiet data (more samples per lane than its share: staging
    passes), loud data, the lane classes of 64 / 128 / 256 lanes per block."""
    import deltarice_amd as dr
    rng = np.random.default_rng(404)
    shapes = [(8, 100, 30011, 3, "gauss10"),      # 800 waveforms, 256 lanes, three blocks each, every block a run of its own
              (4, 1050, 30000, 3, "gauss10"),     # 4200 waveforms: runs of two blocks
              (2, 800, 9001, 3, "gauss10"),       # one block per waveform
              (2, 1600, 6000, 3, "gauss10"),      # 128 lanes per block
              (2, 1600, 3000, 3, "gauss10"),      # 64 lanes
              (3, 300, 20000, 3, "steps"),        # quiet: 4 bits per sample
              (2, 450, 16385, 8, "gauss300"),     # loud under a RiceParameter that suits it
              (1, 801, 25000, 3, "zeros")]
    for si, (n_chunks, W, L, k, kind) in enumerate(shapes):
        N = W * L - (L // 3)  # a shorter last waveform
        x = make_data(rng, kind, n_chunks * N)
        for taps in (TAPS if si == 0 else [TAPS[0], TAPS[3]]):
            opts = (1 << k, L, len(taps)) + u32(taps)
            plan = ctx.plan_uniform(n_chunks, N, opts)
            enc = plan.encode(dev(ctx, x))
            if si in (0, 2):
                ref_w, ref_off = O.encode_batch(x, N, opts)
                w, off = enc.to_numpy()
                assert np.array_equal(off, ref_off) and np.array_equal(w, ref_w), (si, taps)
            outs = {}
            for flags in (0, 2097152, 256):
                ctx.set_option("debug_flags", flags)
                outs[flags] = plan.decode(enc).cpu().numpy()
                path = plan.last_decode_path()
                ctx.set_option("debug_flags", 0)
                if flags == 0:
                    assert path & BLOCKS and path & IIR_FUSED and not path & IIR, (path, W, L)
                elif flags == 2097152:
                    assert path & BLOCKS and path & IIR and not path & IIR_FUSED, (path, W, L)
                else:
                    assert path & LANES_ANY and not path & BLOCKS, (path, W, L)
            for flags, y in outs.items():
                assert np.array_equal(y, x), (n_chunks, W, L, k, kind, taps, flags, int(np.argmax(y != x)))


def test_general_filter_ramp_falls_back_and_is_exact(ctx, O):
    """A slope-1 ramp never lets a speculative parse fall into step: the block decoder flags such waveforms and they are
    decoded again serially -- with the general filter's inverse, not the delta one."""
    import deltarice_amd as dr
    x = make_data(np.random.default_rng(1), "ramp", 2 * 400000)
    for taps in [(1, -1, 1, -1), (-1, 1)]:
        opts = (8, 400000, len(taps)) + u32(taps)
        ref_w, ref_off = O.encode_batch(x, 2 * 400000, opts)
        plan = ctx.plan_uniform(1, 2 * 400000, opts)
        enc = dr.EncodedBatch(dev(ctx, ref_w.view(np.int32)), dev(ctx, ref_off.astype(np.int64)), ref_w.size)
        y = plan.decode(enc).cpu().numpy()
        assert plan.last_decode_path() & BLOCKS
        assert np.array_equal(y, x), taps


def test_ragged_few_long_waveforms_delta_and_general(ctx, O):
    import deltarice_amd as dr
    rng = np.random.default_rng(302)
    Ls = [30000, 70000, 0, 120000, 30000]
    Ns = [30000 * 3 + 7, 70000 * 2, 250000, 120000 + 555, 30000 * 2]
    xs = [rng.normal(0, 10, n).astype(np.int16) for n in Ns]
    x = np.concatenate(xs)
    for taps in [None, (1, -1, 1, -1), (-1, 2, -1)]:
        plan = ctx.plan(Ns, Ls, 8, taps=taps)
        refs = []
        for xc, L in zip(xs, Ls):
            opts = (8, L if L else xc.size) + ((len(taps),) + u32(taps) if taps else ())
            refs.append(O.encode_chunk(xc, opts))
        ref_w = np.concatenate(refs)
        ref_off = np.concatenate([[0], np.cumsum([r.size for r in refs])]).astype(np.int64)
        enc = plan.encode(dev(ctx, x))
        w, off = enc.to_numpy()
        assert np.array_equal(off.astype(np.int64), ref_off) and np.array_equal(w, ref_w), taps
        for flags in (0, 256):
            ctx.set_option("debug_flags", flags)
            y = plan.decode(enc).cpu().numpy()
            path = plan.last_decode_path()
            ctx.set_option("debug_flags", 0)
            assert np.array_equal(y, x), (taps, flags)
            if flags == 0:
                assert path & BLOCKS and (not taps or path & IIR), (path, taps)
            else:
                assert not path & BLOCKS
        # damage that moves code boundaries is reported (or decodes to other values where only remainder bits changed)
        bad = ref_w.copy()
        pos = int(ref_off[1]) + 2 + int(ref_w[int(ref_off[1]) + 1]) // 2
        bad[pos:pos + 3] ^= np.uint32(0x5A5A5A5A)
        encb = dr.EncodedBatch(dev(ctx, bad.view(np.int32)), dev(ctx, ref_off), bad.size)
        try:
            assert plan.decode(encb).cpu().numpy().shape == x.shape
        except dr.DeltaRiceError as e:
            assert e.status == 4
        assert np.array_equal(plan.decode(enc).cpu().numpy(), x)  # the plan is still usable


def test_ragged_batch_with_short_waveforms_keeps_the_lane_decoder(ctx, O):
    # one chunk of short waveforms among long ones: not the block decoder's geometry
    rng = np.random.default_rng(303)
    Ls, Ns = [60000, 512], [60000 * 2, 512 * 30]
    x = rng.normal(0, 10, sum(Ns)).astype(np.int16)
    plan = ctx.plan(Ns, Ls, 8)
    enc = plan.encode(dev(ctx, x))
    assert np.array_equal(plan.decode(enc).cpu().numpy(), x)
    assert not plan.last_decode_path() & BLOCKS
