#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02w2; rm -rf $O; mkdir -p $O; cd $R
for ch in 150 220; do
for f in 0 2048; do
  echo "== len_sweep $ch chunks, flags $f" >> $O/sweep.txt
  DRX_SWEEP_CHUNKS=$ch DRX_DEBUG_FLAGS=$f timeout -k 10 300 python3 tools/len_sweep.py 512 1024 2048 >> $O/sweep.txt 2>&1 || echo "FAILED rc=$?" >> $O/sweep.txt
done; done
grep -v amdgpu.ids $O/sweep.txt
