#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel launch
for the drx:: kernels.  usage: pmc_summary.py counter_collection.csv [more.csv ...]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "drx::" not in name:
                continue
            short = name.split("drx::")[1].split("(")[0]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"  {c:28s} {sum(v) / len(v):18.0f}  (n={len(v)})")
