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
