"""Batch splitter: shards a batch of HDF5 chunks over the ranks of one node.

Chunks are independent filter calls in the reference (src/deltaRice.c:468-490) and
waveforms are independent inside a chunk (:417-426), so the split needs no data-path
collective: rank r takes a contiguous range of the chunk list, encodes/decodes it on
its own GPU, and the only exchange is one all-gather of each rank's encoded size so
that every rank knows the global offset of its part of the encoded stream (the
rank-order concatenation is byte-identical to a single-GPU run).  On GPUs this is
RCCL (torch.distributed backend "nccl") over xGMI; the same code runs on gloo/CPU
tensors for tests.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n_items over world ranks: (first, count); sizes differ by <= 1."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_items, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def shard_table(n_items: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n_items, world, r) for r in range(world)]


def gather_encoded_sizes(local_words: torch.Tensor, group=None) -> torch.Tensor:
    """All-gathers one int64 (this rank's encoded word count); returns int64[world]
    on the same device.  The single collective of the distributed path."""
    world = dist.get_world_size(group)
    lw = local_words.reshape(1).to(torch.int64)
    out = torch.empty(world, dtype=torch.int64, device=lw.device)
    dist.all_gather_into_tensor(out, lw, group=group)
    return out


def global_offsets(sizes: torch.Tensor) -> torch.Tensor:
    """Exclusive prefix of per-rank sizes -> first word of each rank's part (+ total), int64[world+1]."""
    z = torch.zeros(1, dtype=torch.int64, device=sizes.device)
    return torch.cat([z, torch.cumsum(sizes.to(torch.int64), 0)])


def global_chunk_offsets(local_chunk_word_off: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Turns a rank-local chunk offset table into global word offsets.
    Returns (global offsets of this rank's chunks [n_local+1], per-rank sizes [world])."""
    sizes = gather_encoded_sizes(local_chunk_word_off[-1], group)
    base = global_offsets(sizes)[dist.get_rank(group)]
    return local_chunk_word_off + base, sizes


# ---- optional: a batch that lives on ONE rank -----------------------------------------------------------------
# "Optional data movement for a root-resident batch: grouped send/recv from rank 0 to each peer (one xGMI link each, 7 in
# parallel) and the mirror gather.  No all-reduce."  These are not part of bench.py's timed region (every rank makes its
# own shard there); they exist for callers whose raw batch or whose encoded stream has to start / end on one GPU, and
# their time is to be reported separately from the codec's.


def scatter_chunks(x_root, n_chunks: int, chunk_samples: int, device=None, group=None, src: int = 0) -> torch.Tensor:
    """Rank `src` holds the raw batch (int16[n_chunks * chunk_samples]); every rank returns ITS contiguous share of the
    chunk list (shard_range), int16 on `device`.  One grouped batch of point-to-point sends from `src`: with RCCL each goes
    over its own xGMI link."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, count = shard_range(n_chunks, world, rank)
    if rank == src:
        if x_root is None or x_root.numel() != n_chunks * chunk_samples or x_root.dtype != torch.int16:
            raise ValueError("scatter_chunks: the source rank needs the whole int16 batch")
        device = x_root.device
    mine = torch.empty(count * chunk_samples, dtype=torch.int16, device=device)
    ops = []
    if rank == src:
        for r, (f, c) in enumerate(shard_table(n_chunks, world)):
            part = x_root[f * chunk_samples:(f + c) * chunk_samples]
            if r == src:
                mine.copy_(part)
            elif c:
                ops.append(dist.P2POp(dist.isend, part, r, group))
    elif count:
        ops.append(dist.P2POp(dist.irecv, mine, src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return mine


def gather_encoded(words_local: torch.Tensor, n_words_local: int, group=None, dst: int = 0):
    """The mirror: every rank's encoded words (int32 / uint32 storage, the first n_words_local of words_local) end up on
    rank `dst`, concatenated in rank order -- the stream a single GPU would have produced.  Returns (words, offsets[world+1])
    on `dst`, (None, offsets) elsewhere.  One all-gather of sizes, then one grouped batch of receives."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = gather_encoded_sizes(torch.tensor(int(n_words_local), dtype=torch.int64, device=words_local.device), group)
    offs = global_offsets(sizes)
    out = None
    ops = []
    if rank == dst:
        out = torch.empty(int(offs[-1]), dtype=words_local.dtype, device=words_local.device)
        for r in range(world):
            lo, hi = int(offs[r]), int(offs[r + 1])
            if r == dst:
                out[lo:hi].copy_(words_local[:hi - lo])
            elif hi > lo:
                ops.append(dist.P2POp(dist.irecv, out[lo:hi], r, group))
    elif n_words_local:
        ops.append(dist.P2POp(dist.isend, words_local[:int(n_words_local)].contiguous(), dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out, offs
