"""RiceParameter and encoding-filter optimiser (the routine the reference's docs describe but do not ship:
/root/reference/docs/Optimization.md:5-19, "capable of minimizing both in tandem"; docs/Algorithm.md:20).

The search is the one the docs spell out (Optimization.md:17-19): start from the caller's filter; test every filter of the
same length whose taps lie within +-search of the current one (filters with a zero at either end are invalid: the zero changes
nothing but the stored length); if the current filter is the best of its neighbourhood stop, otherwise move to the best and
repeat; sizes already computed are kept.  What is minimised is the EXACT size of the encoded batch -- ``drx_estimate_words``
counts, on the GPU and in one pass per filter, the words the encoder would write for every RiceParameter 2^0 .. 2^15
(``k_estimate_words``, csrc/drx_encode_kernels.hip) -- so M is optimised in tandem: a filter's score is its best k's size.

Nothing here runs on the CPU but the loop over candidate filters; the samples stay in HBM.
"""
from __future__ import annotations

import ctypes as C
import itertools
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from .codec import Context


def _valid(taps: Tuple[int, ...], lossless: bool) -> bool:
    if taps[0] == 0 or taps[-1] == 0:  # Optimization.md:15 ("should not have zeros on either end"); taps[0] divides
        return False
    if lossless and abs(taps[0]) != 1:  # decodeWaveform divides by filter[0] in integers (src/deltaRice.c:91-102): exact
        return False                    # for every input only when it is +-1
    return True


def optimise(ctx: Context, x: torch.Tensor, chunk_samples: int, wave_len: int, taps: Sequence[int] = (1, -1), search: int = 1,
             lossless: bool = True, max_steps: int = 64, ks: Optional[Sequence[int]] = None):
    """-> dict(taps, k, m, words, bits_per_sample, steps, evaluated).

    x: int16 samples on the context's device, a whole number of chunks of chunk_samples; wave_len as in compression_opts
    (<= 0: the whole chunk); taps: the initial filter, its length is kept; search: s of Optimization.md:19 ((2s+1)^n
    neighbours per step); lossless: only filters the reference decodes exactly (taps[0] = +-1); ks: RiceParameters 2^k to
    consider (default all sixteen).  `evaluated` maps every filter tried to its sixteen sizes in words."""
    n = x.numel()
    if n == 0 or n % chunk_samples:
        raise ValueError("x must hold a whole number of chunks")
    n_chunks = n // chunk_samples
    L = wave_len if wave_len and wave_len > 0 else chunk_samples
    ks = list(range(16)) if ks is None else [int(k) for k in ks]
    cur = tuple(int(t) for t in taps)
    if not _valid(cur, lossless):
        raise ValueError(f"initial filter {cur} is not valid (zero at an end" + (", or taps[0] != +-1)" if lossless else ")"))
    seen: Dict[Tuple[int, ...], np.ndarray] = {}
    plan = ctx.plan_uniform(n_chunks, chunk_samples, (8, L))  # (the plan's own k does not enter the estimate)

    def size_of(f: Tuple[int, ...]) -> np.ndarray:
        if f not in seen:
            t = (C.c_int32 * len(f))(*f)
            ctx._check(ctx.lib.drx_plan_set_filter(plan._h, len(f), t))  # ([1, -1] selects the delta kernels again)
            seen[f] = plan.estimate_words(x)
        return seen[f]

    def score(f: Tuple[int, ...]) -> Tuple[int, int]:
        w = size_of(f)
        k = min(ks, key=lambda kk: (int(w[kk]), kk))
        return int(w[k]), k

    steps = 0
    best, best_k = score(cur)
    while steps < max_steps:
        steps += 1
        cand, cand_words, cand_k = cur, best, best_k
        for delta in itertools.product(range(-search, search + 1), repeat=len(cur)):
            f = tuple(t + d for t, d in zip(cur, delta))
            if f == cur or not _valid(f, lossless):
                continue
            w, k = score(f)
            if w < cand_words:
                cand, cand_words, cand_k = f, w, k
        if cand == cur:
            break
        cur, best, best_k = cand, cand_words, cand_k
    plan.close()
    return {"taps": cur, "k": best_k, "m": 1 << best_k, "words": best, "bits_per_sample": 32.0 * best / n,
            "steps": steps, "evaluated": seen}
