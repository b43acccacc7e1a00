#!/usr/bin/env python3
"""Trims a rocprofv3 `--kernel-trace --stats` kernel_stats CSV to one short line per kernel
(torch's template names run to kilobytes).  usage: trim_stats.py in.csv out.csv"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w", newline="") as g:
    r = csv.reader(f)
    w = csv.writer(g)
    for i, row in enumerate(r):
        if i:
            name = row[0]
            row[0] = name if name.startswith(("drx::", "void drx::")) else name[:60].split("<")[0] + " (torch/runtime)"
            row[0] = row[0].split("(drx::Geom")[0]
        w.writerow(row)
