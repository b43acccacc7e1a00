"""One rank of tests/test_gpu_dist.py: encodes its shard of the chunk list on the GPU through the product
path (deltarice_amd.codec -> C ABI -> HIP kernels) inside a torch.distributed process group, exchanges the
encoded sizes (the one collective of the distributed path, deltarice_amd/dist.py) and saves what it produced.

  RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (as torch.distributed.run sets them)
  python tests/dist_worker.py backend n_chunks chunk_samples M L outdir

backend gloo: every rank uses cuda:0 (the rehearsal of the N-GPU flow on a one-GPU box); nccl: cuda:LOCAL_RANK.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    backend, n_chunks, chunk_samples, M, L, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), \
        int(sys.argv[5]), sys.argv[6]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank)) if backend == "nccl" else 0
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    try:
        import deltarice_amd as dr
        from deltarice_amd import dist as drdist
        assert dist.get_world_size() == world
        ctx = dr.Context(local)
        x = np.random.default_rng(99).normal(0, 10, n_chunks * chunk_samples).astype(np.int16)  # same on every rank
        first, count = drdist.shard_range(n_chunks, world, rank)
        if count:
            mine = torch.from_numpy(x[first * chunk_samples:(first + count) * chunk_samples]).to(ctx.device)
            plan = ctx.plan_uniform(count, chunk_samples, (M, L))
            enc = plan.encode(mine)
            words, off = enc.to_numpy()
            local_off = enc.chunk_word_off
            # decode what this rank encoded, on the same stream, inside the same process group
            y = plan.decode(enc).cpu().numpy()
            assert np.array_equal(y, mine.cpu().numpy()), "rank-local round trip failed"
        else:
            words, off = np.zeros(0, np.uint32), np.zeros(1, np.uint64)
            local_off = torch.zeros(1, dtype=torch.int64, device=ctx.device)
        if backend == "nccl":
            with torch.cuda.stream(ctx.stream):
                goff, sizes = drdist.global_chunk_offsets(local_off)
            ctx.stream.synchronize()
        else:
            goff, sizes = drdist.global_chunk_offsets(local_off.cpu())
        assert int(sizes.cpu()[rank]) == words.size
        np.savez(os.path.join(outdir, f"r{rank}.npz"), words=words, goff=goff.cpu().numpy(), first=first, count=count,
                 device=local)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
