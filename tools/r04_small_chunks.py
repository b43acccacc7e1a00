import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import deltarice_amd as dr
ctx = dr.Context(0); ctx.set_option("profile", 1)
for W, n_chunks in ((100, 2000), (64, 3000), (20, 10000), (500, 400), (5000, 40), (8000, 25)):
    L = 7000; N = W * L
    x = (torch.randn(n_chunks * N, device=ctx.device) * 10).to(torch.int16)
    torch.cuda.synchronize()
    plan = ctx.plan_uniform(n_chunks, N, (8, L))
    enc = plan.encode(x)
    for flags in (0, 2048):
        ctx.set_option("debug_flags", flags)
        ts = []
        for _ in range(4):
            y = plan.decode(enc); ts.append(plan.last_timings())
        assert torch.equal(x, y)
        t = np.median(np.array(ts[1:]), axis=0)
        print(f"W={W} chunks={n_chunks} flags={flags}: walk {t[0]:.3f} decode {t[1]:.3f} total {t[3]:.3f} ms path {plan.last_decode_path()}", flush=True)
    ctx.set_option("debug_flags", 0)
    del x, y, enc, plan
