#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02p2; mkdir -p $O; cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces or ragged" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
for ch in 100 5 1; do
for f in 0 4096 32768; do
  echo "== len_sweep $ch chunks, flags $f" >> $O/sweep.txt
  DRX_SWEEP_CHUNKS=$ch DRX_DEBUG_FLAGS=$f timeout -k 10 300 python3 tools/len_sweep.py 512 1024 2048 3500 5000 7000 9000 12000 16384 65536 >> $O/sweep.txt 2>&1 || echo "FAILED rc=$?" >> $O/sweep.txt
done; done
grep -v amdgpu.ids $O/sweep.txt | cut -c1-40
