#!/usr/bin/env python3
"""Reads a DRX_ES_TRACE file (k_encode_stream, -DDRX_ENC_STAMPS build): per waveform five 100 MHz stamps -- 1 begun, 2 size
known, 3 place seen by its coder; in the slot of a ticket's first waveform also 0 the ticket's total published and 4 its place
stored by the scanner.  Prints where a place's delay comes from: the rendezvous (size known -> ticket total published),
in-order completion (an earlier ticket published later), the scanner (all earlier totals published -> place stored), the way
back (place stored -> seen by the coder).  usage: tools/r04_es_trace.py FILE WAVES_PER_TICKET"""
import sys
import numpy as np

wv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 5).astype(np.float64)
n = (len(t) // wv) * wv
t = t[:n]
t0 = t[t > 0].min()
t = np.where(t > 0, (t - t0) / 100.0, np.nan)  # microseconds
beg, known, seen = t[:, 1], t[:, 2], t[:, 3]
tpub, tstored = t[::wv, 0], t[::wv, 4]            # per ticket
front = np.maximum.accumulate(np.nan_to_num(tpub))  # when every total up to this ticket has been published
def q(x):
    x = x[np.isfinite(x)]
    return "mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f  max %8.2f" % (x.mean(), *np.percentile(x, [10, 50, 90, 99]), x.max())
print(f"{n} waveforms, {n // wv} tickets of {wv}; kernel span {np.nanmax(t):.1f} us")
print("begun -> size known                      ", q(known - beg))
print("size known -> ticket's total published   ", q(np.repeat(tpub, wv) - known), " (rendezvous)")
print("total published -> all earlier too       ", q(front - tpub), " (in-order completion)")
print("all earlier published -> place stored    ", q(tstored - front), " (scanner; next ticket's place)")
print("place stored -> seen by the coder        ", q(seen - np.repeat(np.concatenate([[tstored[0]], tstored[:-1]]), wv)), " (the ticket's own place is stored with the ticket before)")
print("size known -> place seen                 ", q(seen - known))
step = max(1, (n // wv) // 20)
print("  ticket      begun     known       pub     front    stored      seen(first waveform)")
for T in range(0, n // wv, step):
    g = T * wv
    print("%8d %10.1f %9.1f %9.1f %9.1f %9.1f %9.1f" % (T, beg[g], known[g], tpub[T], front[T], tstored[T], seen[g]))
