"""CPU: the multi-GPU batch splitter (deltarice_amd/dist.py) on gloo, world_size 2 and 3.

Each rank encodes its shard of the chunk list with the oracle (the CPU stands in for the
rank's GPU); the one collective -- the all-gather of encoded sizes -- gives every rank the
global offset of its part; the rank-order concatenation must be byte-identical to encoding
the whole batch at once."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deltarice_amd import dist as drdist


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 500, 501, 4096):
        for world in (1, 2, 3, 8):
            tab = drdist.shard_table(n, world)
            assert tab[0][0] == 0 and sum(c for _, c in tab) == n
            for (a, ca), (b, _) in zip(tab, tab[1:]):
                assert a + ca == b
            assert max(c for _, c in tab) - min(c for _, c in tab) <= 1
    with pytest.raises(ValueError):
        drdist.shard_range(10, 0, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, chunk_samples, opts, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        x = np.random.default_rng(99).normal(0, 10, n_chunks * chunk_samples).astype(np.int16)  # same on every rank
        first, count = drdist.shard_range(n_chunks, world, rank)
        mine = x[first * chunk_samples:(first + count) * chunk_samples]
        if count:
            words, off = O.encode_batch(mine, chunk_samples, opts)
        else:
            words, off = np.zeros(0, np.uint32), np.zeros(1, np.uint64)
        local_off = torch.from_numpy(off.astype(np.int64))
        goff, sizes = drdist.global_chunk_offsets(local_off)
        assert sizes.tolist()[rank] == words.size
        np.savez(os.path.join(outdir, f"r{rank}.npz"), words=words, goff=goff.numpy(), first=first, count=count)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks", [(2, 6), (3, 7), (2, 1)])
def test_sharded_encode_is_byte_identical(tmp_path, world, n_chunks):
    from oracle import oracle as O
    chunk_samples, opts = 4 * 1000, (8, 1000)
    mp.spawn(_worker, args=(world, _free_port(), n_chunks, chunk_samples, opts, str(tmp_path)), nprocs=world, join=True)
    x = np.random.default_rng(99).normal(0, 10, n_chunks * chunk_samples).astype(np.int16)
    ref_words, ref_off = O.encode_batch(x, chunk_samples, opts)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    cat = np.concatenate([p["words"] for p in parts])
    assert np.array_equal(cat, ref_words), "rank-order concatenation differs from the single-rank stream"
    for p in parts:
        f, c = int(p["first"]), int(p["count"])
        assert np.array_equal(p["goff"], ref_off[f:f + c + 1].astype(np.int64)), "global chunk offsets differ"


def _scatter_worker(rank, world, port, n_chunks, chunk_samples, opts, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        x = np.random.default_rng(5).normal(0, 10, n_chunks * chunk_samples).astype(np.int16)
        src = world - 1  # (not rank 0: the helpers take the root as an argument)
        mine = drdist.scatter_chunks(torch.from_numpy(x) if rank == src else None, n_chunks, chunk_samples, device="cpu", src=src)
        first, count = drdist.shard_range(n_chunks, world, rank)
        assert np.array_equal(mine.numpy(), x[first * chunk_samples:(first + count) * chunk_samples]), "scatter: wrong share"
        if count:
            words, _ = O.encode_batch(mine.numpy(), chunk_samples, opts)
        else:
            words = np.zeros(0, np.uint32)
        buf = torch.zeros(words.size + 17, dtype=torch.int32)  # (an output buffer is larger than what was encoded)
        buf[:words.size] = torch.from_numpy(words.view(np.int32))
        out, offs = drdist.gather_encoded(buf, words.size, dst=0)
        assert int(offs[rank + 1] - offs[rank]) == words.size
        if rank == 0:
            np.save(os.path.join(outdir, "gathered.npy"), out.numpy().view(np.uint32))
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks", [(2, 5), (3, 7), (3, 2)])
def test_root_resident_batch_scatter_and_gather(tmp_path, world, n_chunks):
    """SURVEY 8e's optional data movement: the raw batch starts on one rank, the encoded stream ends on one rank; what arrives
    is the stream a single rank would have produced."""
    from oracle import oracle as O
    chunk_samples, opts = 3 * 1000, (8, 1000)
    mp.spawn(_scatter_worker, args=(world, _free_port(), n_chunks, chunk_samples, opts, str(tmp_path)), nprocs=world, join=True)
    x = np.random.default_rng(5).normal(0, 10, n_chunks * chunk_samples).astype(np.int16)
    ref_words, _ = O.encode_batch(x, chunk_samples, opts)
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), ref_words)


def test_bench_gpus_n_without_devices_fails_loudly():
    """`python bench.py --gpus N` is the driver's multi-GPU command: with fewer than N devices it must exit
    non-zero and print no JSON line (it used to run one rank and report n_gpus: 1)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has the GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cpu-seconds", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "visible GPUs" in r.stderr + r.stdout
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
