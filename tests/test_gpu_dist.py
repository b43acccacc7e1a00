"""GPU: the multi-rank flow through the HIP path.

Two ranks in a torch.distributed group encode their shards of a chunk list with deltarice_amd (C ABI -> HIP
kernels) and exchange encoded sizes; the rank-order concatenation must equal, byte for byte, both a one-plan
GPU encode of the whole list and the oracle's bytes (what is sharded is the independent-chunk /
independent-waveform loop of src/deltaRice.c:417-426).  On a one-GPU box the ranks share cuda:0 and talk over
gloo -- the rehearsal of the RCCL flow; with two or more GPUs visible the same test also runs over nccl.

Also: `python bench.py --gpus N` must start and verify N ranks by itself (the driver's command line)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(backend, world, args, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), backend, *map(str, args)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"


def _backends():
    b = [("gloo", 2), ("gloo", 3)]
    if torch.cuda.device_count() >= 2:
        b.append(("nccl", 2))
    return b


@pytest.mark.parametrize("backend,world", _backends())
@pytest.mark.parametrize("n_chunks", [7, 1])
def test_sharded_hip_encode_is_byte_identical(tmp_path, backend, world, n_chunks):
    from oracle import oracle as O
    import deltarice_amd as dr
    chunk_samples, opts = 40 * 3500, (8, 3500)
    _run_ranks(backend, world, (n_chunks, chunk_samples, opts[0], opts[1], tmp_path))
    x = np.random.default_rng(99).normal(0, 10, n_chunks * chunk_samples).astype(np.int16)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    cat = np.concatenate([p["words"] for p in parts])
    # ... equals a single-plan GPU encode of the whole chunk list
    ctx = dr.Context(0)
    enc = ctx.plan_uniform(n_chunks, chunk_samples, opts).encode(torch.from_numpy(x).to(ctx.device))
    one_words, one_off = enc.to_numpy()
    assert np.array_equal(cat, one_words), "rank-order concatenation differs from the one-GPU stream"
    # ... and the oracle's bytes
    ref_words, ref_off = O.encode_batch(x, chunk_samples, opts)
    assert np.array_equal(cat, ref_words) and np.array_equal(one_off, ref_off)
    for p in parts:
        f, c = int(p["first"]), int(p["count"])
        assert np.array_equal(p["goff"], ref_off[f:f + c + 1].astype(np.int64)), "global chunk offsets differ"
    if backend == "nccl":
        assert sorted(int(p["device"]) for p in parts) == list(range(world))


def test_bench_starts_its_own_ranks(tmp_path):
    # the driver's command line, with no launcher around it; two ranks on the one visible GPU over gloo
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--waves", "4000", "--steps", "2",
           "--warmup", "1", "--cpu-seconds", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and len(res["rank_encoded_bytes"]) == 2
    assert all(b > 0 for b in res["rank_encoded_bytes"]) and res["scaling"] == "weak"
    assert 0.39 < res["compression_ratio"] < 0.42


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--cpu-seconds", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible GPUs" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines()), "no JSON line may be printed"
