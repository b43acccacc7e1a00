#!/usr/bin/env python3
"""File <-> VRAM rates of the direct-chunk path vs the per-chunk filter callback, in the units of the
reference's docs/Performance.md (MB/s of uncompressed data, file on tmpfs).
usage: h5_direct_bench.py [rows=20000] [cols=7000] [chunk_rows=2000]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deltarice_amd as dr  # noqa: E402
from deltarice_amd import h5io  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
crows = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
HDF5_DIR = os.environ.get("HDF5_DIR", "/opt/conda")
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
exe = os.path.join(tempfile.mkdtemp(), "h5_roundtrip")  # /dev/shm may be mounted noexec
subprocess.run(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tests", "h5_roundtrip.c"), f"-I{HDF5_DIR}/include",
                f"-L{HDF5_DIR}/lib", "-lhdf5", f"-Wl,-rpath,{HDF5_DIR}/lib"], check=True)
env = dict(os.environ, HDF5_PLUGIN_PATH=os.path.join(ROOT, "deltarice_amd", "plugin"))
ctx = dr.Context(0)
x = (torch.randn(rows * cols, device=ctx.device) * 10).to(torch.int16)
mb = rows * cols * 2 / 1e6
f1 = os.path.join(tmp, "direct.h5")
for _ in range(2):
    t = time.perf_counter(); st_w = h5io.write(ctx, f1, "test", x, rows, cols, crows, 8, cols); tw = time.perf_counter() - t
y = torch.empty_like(x)
for _ in range(2):
    t = time.perf_counter(); st_r = h5io.read(ctx, f1, "test", y); tr = time.perf_counter() - t
assert torch.equal(x, y)
print(f"dataset {rows}x{cols} int16 = {mb:.0f} MB, chunks {crows}x{cols}, stored {st_w['stored_bytes'] / 1e6:.0f} MB")
print(f"direct  VRAM->file {mb / tw:8.0f} MB/s  (gpu {st_w['t_gpu'] * 1e3:.1f} ms, pcie {st_w['t_pcie'] * 1e3:.1f} ms, hdf5 {st_w['t_file'] * 1e3:.1f} ms)")
print(f"direct  file->VRAM {mb / tr:8.0f} MB/s  (hdf5 {st_r['t_file'] * 1e3:.1f} ms, pcie {st_r['t_pcie'] * 1e3:.1f} ms, gpu {st_r['t_gpu'] * 1e3:.1f} ms)")
raw, f2, back = os.path.join(tmp, "raw.bin"), os.path.join(tmp, "filter.h5"), os.path.join(tmp, "back.bin")
x.cpu().numpy().tofile(raw)
t = time.perf_counter(); subprocess.run([exe, "write", f2, raw, str(rows), str(cols), str(crows), "8", str(cols)], env=env, check=True); tw2 = time.perf_counter() - t
t = time.perf_counter(); subprocess.run([exe, "read", f2, back], env=env, check=True); tr2 = time.perf_counter() - t
print(f"filter  RAM->file  {mb / tw2:8.0f} MB/s  (whole process: start-up, raw file read, H5Dwrite through the H5Z callback)")
print(f"filter  file->RAM  {mb / tr2:8.0f} MB/s  (whole process: H5Dread through the H5Z callback, raw file write)")
