import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import deltarice_amd as dr
ctx = dr.Context(0)
rng = np.random.default_rng(404)
n_chunks, W, L, k = 8, 100, 30011, 3
N = W * L - (L // 3)
x = rng.normal(0, 10, n_chunks * N).astype(np.int16)
taps = (1, -1, 1, -1)
opts = (1 << k, L, len(taps)) + tuple(t & 0xFFFFFFFF for t in taps)
plan = ctx.plan_uniform(n_chunks, N, opts)
xd = torch.from_numpy(x).to(ctx.device)
enc = plan.encode(xd)
for flags in (0, 2097152):
    ctx.set_option("debug_flags", flags)
    y = plan.decode(enc).cpu().numpy()
    print("flags", flags, "path", plan.last_decode_path(), "equal", np.array_equal(y, x))
    bad = np.nonzero(y != x)[0]
    if bad.size:
        print(" mismatches", bad.size, "first", bad[:5], "last", bad[-3:])
        # per waveform: first bad sample index within waveform
        wave = []
        for c in range(n_chunks):
            for w in range(W):
                s0 = c * N + w * L
                s1 = min(s0 + L, (c + 1) * N)
                b = np.nonzero(y[s0:s1] != x[s0:s1])[0]
                if b.size: wave.append((c, w, int(b[0]), int(b.size), s1 - s0))
        print(" bad waveforms", len(wave), wave[:12])
        c, w, i0, nb, ln = wave[0]
        s0 = c * N + w * L
        print(" around first:", i0, y[s0 + i0 - 4:s0 + i0 + 8], x[s0 + i0 - 4:s0 + i0 + 8])
