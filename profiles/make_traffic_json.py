#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, collected in separate passes as
MI355X_MICROARCH.md prescribes) into per-launch HBM traffic of the drx:: kernels.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests at
64 B, i.e. reports half of a wide streaming read -- calibrated here against TCC_EA0_RDREQ x 128 B
and against the known size of the stream each kernel reads exactly once; WRITE_SIZE is exact for
16-byte-per-lane streaming stores.  Both are in KB.

usage: make_traffic_json.py out.json counter_collection.csv [...]     (DRX_TRAFFIC_SOURCE: the "source" string)
bench.py imports summarise() / kernel_sources_sha() for its own in-run collection (--collect-traffic)."""
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha():
    """Hash of the kernel sources: a traffic file carries it so that a reader can tell whether the kernels changed since."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "deltarice_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "deltarice_amd", "csrc", "*.h"))):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def summarise(csv_paths):
    acc = defaultdict(lambda: defaultdict(list))
    for path in csv_paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if "drx::" not in name:
                    continue
                short = name.split("drx::")[1].split("(")[0].split("<")[0]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        e = {"launches_averaged": min(len(v) for v in c.values())}
        if "FETCH_SIZE" in m:
            e["hbm_read_bytes"] = 2.0 * m["FETCH_SIZE"] * 1024.0
            if "TCC_EA0_RDREQ_sum" in m:
                e["rdreq_x128B"] = m["TCC_EA0_RDREQ_sum"] * 128.0
        if "WRITE_SIZE" in m:
            e["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024.0
        if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        e["raw_counters"] = m
        out[k] = e
    return out


def main():
    out = summarise(sys.argv[2:])
    json.dump({"source": os.environ.get("DRX_TRAFFIC_SOURCE", "rocprofv3 --pmc, bench.py default workload (1M x 7000, m=8), one MI355X"),
               "git_head": git_head(), "kernel_sources_sha16": kernel_sources_sha(), "kernels": out}, open(sys.argv[1], "w"), indent=1)
    print(json.dumps({k: {a: b for a, b in v.items() if a != "raw_counters"} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
